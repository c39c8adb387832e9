/*
 * acarsdec_amd.h -- C ABI of the MI355X-native acarsdec DSP hot path.
 *
 * This is the drop-in boundary: a plain-C shared library (libacarsdec_amd.so) that a C host
 * (the reference's acarsdec, or any FFI) binds instead of the reference's per-channel DSP
 *      rtl.c:314-361  in_callback()   u8 I/Q -> NCO mix + decimate -> |D| -> dm_buffer
 *      msk.c:67-137   demodMSK()      MSK matched filter / bit-clock PLL -> putbit()
 *      msk.c:53-63    putbit()  ->  acars.c:246-375 decodeAcars()   (framing FSM, fed back into the PLL)
 * All citations are file:line into TLeconte/acarsdec v3.7.  No torch / C++ types cross this
 * boundary: pointers, sizes, ints.  Every function returns ACG_OK (0) or a negative ACG_E* code;
 * nothing throws.  One caller thread per context (same rule as the reference: all DSP runs on the
 * SDR callback thread, rtl.c:363-364).
 *
 * Two layers:
 *   1. batched API (acg_*, this header): thousands of channels per GPU, state resident in HBM across calls;
 *   2. legacy view (compat_msk.c, built inside the reference tree; entry points: acarsdec_amd_compat.h):
 *      initMsk()/demodMSK() with the reference's own signatures (acarsdec.h:190-191) on top of (1), so
 *      acars.c/output.c link unchanged.  See INTEGRATION.md.
 * Measurement and diagnostic entry points (tuning switches, probes, self tests, generators) are NOT product API: they are
 * declared in acarsdec_amd_lab.h.  The shared library exports exactly what THIS header and acarsdec_amd_lab.h declare (a linker
 * version script made from them); what acarsdec_amd_compat.h declares lives in compat_msk.c, which is compiled into the
 * reference program, not into the library.
 */
#ifndef ACARSDEC_AMD_H
#define ACARSDEC_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACG_OK          0
#define ACG_EINVAL     -1   /* bad argument / configuration */
#define ACG_ENOMEM     -2   /* host or device allocation failed */
#define ACG_EHIP       -3   /* HIP runtime error (acg_last_error has the text) */
#define ACG_ENODEV     -4   /* no usable GPU: the library has NO CPU fallback */
#define ACG_EOVERFLOW  -5   /* results were LOST: the device lapped the block queue (the host collected too rarely), or a per-bit
                               log was longer than its buffer; what could be handed out has been.  From acg_drain_* /
                               acg_collect_* it reports the loss ONCE and may still leave blocks queued (the caller's buffer was
                               smaller than what survived): call again -- the next call returns ACG_OK or ACG_EAGAIN */
#define ACG_ESTATE     -6   /* call sequence error */
#define ACG_EAGAIN     -7   /* drain/collect: the caller's buffer is full and more results are queued -- nothing is lost,
                               call again */

#define ACG_INTRATE     12500     /* acarsdec.h:31 */
#define ACG_BLOCK       1024      /* rtl.c:49 RTLOUTBUFSZ: outputs per reference callback */
#define ACG_MAXDECIM    320       /* rtl.c:39 RTLMULTMAX: limit of the u8 I/Q path */
#define ACG_MAXDECIM_SAMPLES 1024 /* limit of the acg_*_samples_* formats (air.c:213: 10 Msps -> 800) */
#define ACG_FLEN        11        /* msk.c:25 */
#define ACG_TXTMAX      250       /* acarsdec.h:55 */

/* acg_config.flags */
#define ACG_F_BITLOG    1u        /* keep per-bit {soft symbol, level} records of each call (8 B per bit to device memory: 2 % of the
                                   * demodulator's time at 1024 channels, 5 % at 16 384; a host that only wants messages leaves it off) */
#define ACG_F_TIMING    2u        /* bracket kernels with HIP events (acg_get_timing) */
#define ACG_F_REPAIR    4u        /* run the block thread's check/repair (acars.c:93-215) on the device:
                                     drain/collect then return what outputmsg() receives -- parity/CRC
                                     verified or repaired, parity stripped, err = parity errors found --
                                     and omit the blocks the reference drops */
#define ACG_F_EXACT_FIR 8u        /* verification mode: the u8 down-converter adds the rtlMult terms of rtl.c:349-351 one after
                                     the other, products and 127.37 per sample rounded separately -- what an IEEE (-O2) build of
                                     the reference executes -- so dm and everything after it is BIT-IDENTICAL to that build.
                                     The streaming kernels re-associate the sum (|d dm| <= 1e-5 |dm|, like the reference's own
                                     -Ofast build); this one is ~20x slower and exists to prove that nothing else differs */

#define ACG_F_PRECISE_MIXER 16u   /* verification mode: the demodulator's mixer (msk.c:86-91, cexp) evaluates sin/cos with the < 1 ulp
                                     polynomial instead of the product's table + rotation (<= 2.1 ulp).  The loop keeps only the
                                     float-rounded products in*cos, in*(-sin), which are the same for both: this flag lets a
                                     maintainer check that on his own input (every bit, state double and block must not change).
                                     ~10 % slower per bit */

typedef struct acg_ctx acg_ctx;

typedef struct {
	int device;         /* HIP device ordinal */
	int nch;            /* channels */
	int nstreams;       /* I/Q streams: nch (one stream per channel) or fewer (rtl.c shape: all
	                       channels of a dongle share one stream, rtl.c:344-354) */
	int decim;          /* M = rtlMult (rtl.c:35-37): input rate = 12500*M */
	int ntaps;          /* complex taps per channel, 1..decim.  Reference: ntaps == decim */
	int max_blocks;     /* capacity: 1024-output blocks per acg_process_* call */
	uint32_t flags;
	int max_lag;        /* most process calls acg_collect_* may stay behind (sizes the block queue: the worst case of
	                       max_lag + 1 calls, 304 B per block, rounded UP to a power of two -- so up to twice that worst
	                       case is allocated; acg_max_lag() says what the queue really holds).  0 = as many calls as fit
	                       into a power-of-two queue of at most 512 MiB, at most 6 (very wide contexts: two calls,
	                       whatever they cost); a host that collects with lag <= 1 after every call asks for 1 */
} acg_config;

/* MSK + framing fields of channel_t (acarsdec.h:76-89) */
typedef struct {
	double MskPhi, MskDf, MskLvlSum;
	float MskClk;
	int MskBitCount;
	unsigned int MskS, idx;
	float inb[2 * ACG_FLEN];      /* re,im interleaved (acarsdec.h:83) */
	int outbits, nbits, Acarsstate;
	int blk_len, blk_err;
	int soh_back;                 /* not in channel_t: 12.5 kHz samples since the SOH byte of the block being assembled completed
	                                 (where acars.c:290 stamps blk->tv; what soh_sample of that block will be measured from).
	                                 Meaningful while Acarsstate is TXT / CRC1 / CRC2, else 0.  acg_set_state() re-bases it on the
	                                 destination slot's own sample counter, so a channel moved mid-block keeps its time stamp;
	                                 a host that leaves it 0 stamps the block at the moment of the move. */
} acg_chan_state;

/* A message block as decodeAcars() queues it (msgblk_t, acarsdec.h:48-57; acars.c:350-364):
 * text still carries parity bits, no repair applied. */
typedef struct {
	int chn;
	int len;
	int err;
	float lvl;                    /* 10*log10(MskLvlSum/MskBitCount), acars.c:351 */
	unsigned char crc[2];
	unsigned char txt[ACG_TXTMAX];
	long long end_bit;            /* per-channel index of the bit that completed the block */
	long long end_sample;         /* per-channel 12.5 kHz sample index of that bit */
	long long soh_sample;         /* per-channel 12.5 kHz sample index of the bit that completed the block's SOH byte: the
	                                 moment the reference stamps blk->tv (acars.c:290 gettimeofday), which every printer and
	                                 JSON sink reports (output.c:162,227,244,361).  Epoch rule: a host that noted the wall
	                                 clock t0 of the channel's first sample since acg_reset sets
	                                     tv = t0 + soh_sample / 12500 s
	                                 (INTEGRATION.md "time stamps"); soh_sample <= end_sample, both count from acg_reset */
} acg_frame;

/* A message as outputmsg() splits it out of a processed block (acarsmsg_t, acarsdec.h:108-124; output.c:486-560,
 * 566-568,623-631 in the build without libacars): the fixed binary record every sink (printmsg, buildjson, Netout*)
 * formats from.  Strings are NUL terminated like the reference's; txt is txt_len bytes, not terminated.  The CLI's
 * filters (-A airflt, label_filter) are not applied. */
#define ACG_MSGTXTMAX   242
typedef struct {
	int chn;
	int err;                      /* parity errors repaired (acars.c:156) */
	float lvl;                    /* dB, acars.c:351 */
	int txt_len;
	long long end_bit, end_sample;   /* as in acg_frame */
	long long soh_sample;         /* as in acg_frame: where blk->tv is taken (acars.c:290); msg->tv = t0 + soh_sample / 12500 s */
	int reserved1;
	char reserved2;
	char mode;
	char addr[8];                 /* aircraft registration without the leading dots */
	char ack;                     /* NAK is reported as '!' (output.c:511-514) */
	char label[3];                /* a DEL second character is reported as 'd' (output.c:518-520) */
	char bid;                     /* block id; '0'..'9' = downlink (output.c:31) */
	char no[5];                   /* message number, downlinks only */
	char fid[7];                  /* flight id, downlinks only */
	char bs, be;                  /* start / end of text characters (STX or ETX / ETX or ETB) */
	char down;
	char txt[ACG_MSGTXTMAX];
	int reserved3;
} acg_msg;

/* ---- lifetime ---------------------------------------------------------------------------- */
int  acg_device_count(void);
int  acg_create(acg_ctx **out, const acg_config *cfg);
void acg_destroy(acg_ctx *ctx);
const char *acg_strerror(int code);
const char *acg_last_error(const acg_ctx *ctx);
const char *acg_version(void);

/* ---- channel set-up (host side of rtl.c:268-287) ------------------------------------------ */
/* rtl.c:131-168 chooseFc(): Fd (Hz) is sorted in place; returns Fc or 0 ("too far apart"). */
unsigned int acg_rtl_choose_fc(unsigned int *Fd, unsigned int nbch, int decim);
/* rtl.c:283-286: wf[ind] = cexpf(-j*AMFreq*ind)/rtlMult/127.5, taps_out is [decim][2]. */
int  acg_rtl_taps(int Fr_hz, unsigned int Fc_hz, int decim, float *taps_out);
/* taps: [n][ntaps][2] float for channels ch0..ch0+n-1 */
int  acg_set_taps(acg_ctx *ctx, int ch0, int n, const float *taps);
/* stream index of every channel; default: channel c reads stream c % nstreams */
int  acg_set_channel_streams(acg_ctx *ctx, const int *stream_of_channel);
/* initMsk() (msk.c:30-51) + initAcars() (acars.c:230-234) for all channels */
int  acg_reset(acg_ctx *ctx);

/* ---- the hot path ------------------------------------------------------------------------- */
/* in_callback() for every channel: iq is [nstreams] rows of interleaved u8 I,Q, row r at
 * iq + r*pitch_bytes, each nblocks*1024*decim*2 bytes (rtl.c:330).  *_dev: device pointer,
 * asynchronous: the down-converter (the only reader of iq) is enqueued on hip_stream (NULL = the
 * context's own stream) so later work on that stream may overwrite iq; the demodulator follows on
 * an internal stream.  Results are complete after acg_sync / acg_drain_frames / acg_collect_frames.
 * *_host: the librtlsdr contract (rtl.c:314-330: the buffer is the driver's again when the callback returns) -- the call
 * returns as soon as the input has LEFT iq_host; it went to one of two device staging buffers on a copy stream of its own,
 * beside the kernels of the previous call, and this call's kernels run after the return.  From pinned memory
 * (acg_host_alloc, or the caller's own buffers through acg_host_register) that copy is one DMA at the link's rate; pageable
 * memory works and is staged by the runtime. */
int  acg_process_iq_u8_dev(acg_ctx *ctx, const uint8_t *iq_dev, size_t pitch_bytes, int nblocks,
			   void *hip_stream);
int  acg_process_iq_u8_host(acg_ctx *ctx, const uint8_t *iq_host, size_t pitch_bytes, int nblocks);
/* demodMSK() for every channel straight from 12.5 kHz samples (soundfile.c:71-77, alsa.c:122):
 * dm is [nch] rows of `len` floats, row c at dm + c*pitch_floats; len <= max_blocks*1024.
 * *_dev: asynchronous like the iq entry points -- dm is consumed after what hip_stream holds so far, and
 * later work on hip_stream (refilling dm in place) is ordered behind the demodulator that reads it. */
int  acg_process_dm_dev(acg_ctx *ctx, const float *dm_dev, size_t pitch_floats, int len,
			void *hip_stream);
int  acg_process_dm_host(acg_ctx *ctx, const float *dm_host, size_t pitch_floats, int len);
/* only the down-converter (rtl.c:332-354), no demodMSK: leaves dm readable by acg_read_dm */
int  acg_fir_only_dev(acg_ctx *ctx, const uint8_t *iq_dev, size_t pitch_bytes, int nblocks,
		      void *hip_stream);
int  acg_sync(acg_ctx *ctx);

/* ---- the other front ends' sample formats ----------------------------------- */
#define ACG_FMT_CS16       1   /* interleaved int16 I,Q: soapy.c:238-241 (dm = |D| with the /32768 of soapy.c:241) */
#define ACG_FMT_S16_SPLIT  2   /* int16 I plane + int16 Q plane: sdrplay.c:219-225 (dm = |D|/4) */
#define ACG_FMT_F32_REAL   3   /* real float32 samples, complex taps: air.c:314-324 */
/* soapy.c:163-166 oscillator table ([decim][2]); ch->Fr is a float in that front end */
int  acg_soapy_taps(float Fr_hz, int freq_hz, int decim, float *taps_out);
/* sdrplay.c:160-164 (fixed rate multiplier 160) */
int  acg_sdrplay_taps(float Fr_hz, unsigned int Fc_hz, float *taps_out);
/* air.c:62 centre frequency (no IF-filter branch) and air.c:278-285 taps; inrate = 12500*decim */
unsigned int acg_airspy_choose_fc(unsigned int minF_hz, unsigned int maxF_hz);
int  acg_airspy_taps(int Fr_hz, int Fc_hz, unsigned int inrate, float *taps_out);
/* Window-aligned device input: [nstreams] rows of nblocks*1024*decim samples (4 bytes per sample;
 * ACG_FMT_S16_SPLIT: I plane at the row start, Q plane plane_bytes further).  decim % 4 == 0
 * and decim <= ACG_MAXDECIM_SAMPLES (split planes: decim % 8 == 0, decim <= 208).  Same asynchronous
 * contract and FIR/demodulator stream pipeline as acg_process_iq_u8_dev. */
int  acg_process_samples_dev(acg_ctx *ctx, int fmt, const void *dev, size_t pitch_bytes, size_t plane_bytes,
			     int nblocks, void *hip_stream);
/* Host input of ANY length per call, as the SDR drivers deliver it: windows may straddle calls (the
 * reference carries D / its index across buffers: soapy.c:232-254, sdrplay.c:215-236, air.c:299-338).
 * p0 = samples (I plane for ACG_FMT_S16_SPLIT), p1 = Q plane or NULL; rows pitch_samples apart.  Same return contract
 * and the same two staging buffers as acg_process_iq_u8_host: the copy of feed i+1 runs beside the kernels of feed i. */
int  acg_feed_samples_host(acg_ctx *ctx, int fmt, const void *p0, const void *p1, size_t pitch_samples,
			   size_t nsamples);

/* pinned host memory for the *_host entry points (hipHostMalloc / hipHostRegister behind a C face, so that a C host needs
 * no HIP headers): acg_host_alloc returns NULL on failure */
void *acg_host_alloc(size_t bytes);
void acg_host_free(void *p);
int  acg_host_register(void *p, size_t bytes);
int  acg_host_unregister(void *p);

/* ---- results ------------------------------------------------------------------------------ */
/* Blocks completed since the last drain/collect, ordered by (chn, end_bit) within the call.  Waits for ALL
 * enqueued work of the context.  If more blocks are queued than max_frames, the oldest max_frames are handed out, the
 * rest STAY queued and the call returns ACG_EAGAIN (call again: across calls the order is completion order). */
int  acg_drain_frames(acg_ctx *ctx, acg_frame *out, int max_frames, int *nframes);
/* Streaming variant: hands over the blocks of every process call except the `lag` most recent
 * ones and waits only for those older calls, so that the newest call(s) keep the GPU busy
 * (lag = 1: classic double buffering; lag = 0: wait for the last call only).  0 <= lag <= acg_max_lag(). */
int  acg_collect_frames(acg_ctx *ctx, int lag, acg_frame *out, int max_frames, int *nframes);
/* Largest lag this context accepts: its block queue is sized for the worst case (a 56-bit block every 291 samples on
 * every channel) of acg_max_lag() + 1 calls, so a host that collects after every call cannot be lapped.
 * At least acg_config.max_lag if that was given (the power-of-two queue may hold more), else 6 unless that would take
 * more than 512 MiB (then fewer, at least 1). */
int  acg_max_lag(const acg_ctx *ctx);
/* SURVEY 8f.4, the batch sink: like acg_drain_frames / acg_collect_frames, but every block is taken through
 * outputmsg()'s field split on the device and handed over as a fixed binary record.  Needs ACG_F_REPAIR (outputmsg()
 * receives repaired, parity-stripped blocks); blocks the repair drops are omitted.  Ordered by (chn, end_bit) within the
 * call.  A call looks at the oldest max_msgs queued blocks only (splits, copies and consumes exactly those); if more are
 * queued they STAY queued and the call returns ACG_EAGAIN ("call again"; across calls the order is completion order).
 * Every byte of a record is defined (unused text bytes are 0). */
int  acg_drain_msgs(acg_ctx *ctx, acg_msg *out, int max_msgs, int *nmsgs);
int  acg_collect_msgs(acg_ctx *ctx, int lag, acg_msg *out, int max_msgs, int *nmsgs);
/* Per-bit records of the LAST process call for one channel (needs ACG_F_BITLOG):
 * vo = the value putbit() receives (msk.c:122-126), lvl = cabsf(v) (msk.c:110). */
int  acg_read_bits(acg_ctx *ctx, int ch, float *vo, float *lvl, int max_bits, int *nbits);
/* all channels at once: counts[nch], vo/lvl [nch][cap] with cap = acg_bit_capacity() */
int  acg_read_bits_all(acg_ctx *ctx, int *counts, float *vo, float *lvl);
int  acg_bit_capacity(const acg_ctx *ctx);
/* dm_buffer of the last call (rtl.c:353), n floats of channel ch */
int  acg_read_dm(acg_ctx *ctx, int ch, float *dm, int n);
int  acg_get_state(acg_ctx *ctx, int ch, acg_chan_state *st);
int  acg_set_state(acg_ctx *ctx, int ch, const acg_chan_state *st);     /* ACG_EINVAL: idx >= 11, blk_len outside 0..241 */
/* the same for channels ch0 .. ch0+n-1 in ONE transfer each way (the legacy view moves all of a dongle's channels per
 * callback: rtl.c:344-360), and dm_buffer of the last call for n channels: row i at dm + i*pitch_floats, nfloats each */
int  acg_get_state_n(acg_ctx *ctx, int ch0, int n, acg_chan_state *st);
int  acg_set_state_n(acg_ctx *ctx, int ch0, int n, const acg_chan_state *st);
int  acg_read_dm_n(acg_ctx *ctx, int ch0, int n, float *dm, size_t pitch_floats, int nfloats);
/* blk->txt of the block channel ch is assembling (acars.c:304 appends to it; blk_len bytes of it are meaningful): the part of
 * channel_t's state that is not a scalar.  A host that moves a channel between slots or contexts in the middle of a block takes
 * it along with acg_get_state / acg_set_state (the legacy view does not need it: there the text lives in the caller's ch->blk).
 * txt: ACG_TXTMAX bytes. */
int  acg_get_block_text(acg_ctx *ctx, int ch, unsigned char *txt);
int  acg_set_block_text(acg_ctx *ctx, int ch, const unsigned char *txt);

/* Replays the bit records of the last call through a putbit()-shaped sink, channel by channel
 * in channel order: for every bit sink(user, ch, vo, lvl).  The legacy shim's sink performs
 * msk.c:112-113 + putbit() on the caller's channel_t, i.e. calls the UNCHANGED decodeAcars(). */
typedef void (*acg_bit_sink)(void *user, int ch, float vo, float lvl);
int  acg_replay_bits(acg_ctx *ctx, acg_bit_sink sink, void *user);

/* ---- timing ------------------------------------------------------------------------------- */
/* Sums of HIP-event-bracketed kernel time since the last call (ACG_F_TIMING), in ms, and the
 * number of launches they cover.  Synchronises. */
int  acg_get_timing(acg_ctx *ctx, double *fir_ms, int *fir_launches, double *msk_ms, int *msk_launches);
/* 0 = no events, 1 = both stages (what ACG_F_TIMING starts with), 2 = down-converter launches only:
 * event records on the demodulator stream sit on its serial launch chain (~10 us per launch). */
int  acg_set_timing(acg_ctx *ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif /* ACARSDEC_AMD_H */
