/*
 * acars_oracle.h -- CPU restatement of the acarsdec per-channel DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker.  The product path
 * (acarsdec_amd/) never links, imports or calls it.
 *
 * Parity pinning: the reference ships no golden vectors of its own (no tests
 * at all).  This restatement is pinned against (1) the unmodified reference
 * sources compiled by oracle/Makefile into oracle/_ref/ (bit-identical state,
 * bits and frames on test.wav and on synthetic IQ, see tests/test_oracle_vs_ref.py)
 * and (2) the committed fixtures under tests/golden/ generated from that build.
 *
 * All file:line citations are into the reference tree (TLeconte/acarsdec v3.7).
 */
#ifndef ACARS_ORACLE_H
#define ACARS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INTRATE 12500          /* acarsdec.h:31 */
#define ORC_FLEN 11                /* msk.c:25  (INTRATE/1200)+1 */
#define ORC_MFLTOVER 12            /* msk.c:26 */
#define ORC_FLENO 133              /* msk.c:27  FLEN*MFLTOVER+1 */
#define ORC_TXTMAX 250             /* acarsdec.h:55 */

/* frame-FSM states, acarsdec.h:88 */
enum { ORC_WSYN = 0, ORC_SYN2, ORC_SOH1, ORC_TXT, ORC_CRC1, ORC_CRC2, ORC_END };

/* a message block as it reaches the block queue (acarsdec.h:48-57, acars.c:350-364) */
typedef struct {
	int chn;
	int len;
	int err;
	float lvl;
	unsigned char crc[2];
	unsigned char txt[ORC_TXTMAX];
	long long end_bit;      /* index (per channel, from 0) of the bit that completed the block */
	long long end_sample;   /* index (per channel, from 0) of the 12.5 kHz sample whose bit completed the block */
	long long soh_sample;   /* ... whose bit completed the block's SOH byte: where acars.c:290 takes blk->tv */
} orc_frame;

/* per-bit log entry (what putbit() receives + the level, msk.c:110-126) */
typedef struct {
	float vo;               /* signed soft symbol handed to putbit (after the MskS&2 polarity) */
	float lvl;              /* cabsf(v) before normalisation */
} orc_bit;

/* per-channel state: the MSK + framing fields of channel_t (acarsdec.h:76-89) */
typedef struct {
	int chn;
	double MskPhi;
	double MskDf;
	float MskClk;
	double MskLvlSum;
	int MskBitCount;
	unsigned int MskS, idx;
	float inb[2 * ORC_FLEN];        /* re,im interleaved */
	unsigned char outbits;
	int nbits;
	int Acarsstate;
	int blk_len, blk_err;
	unsigned char blk_txt[ORC_TXTMAX + 6];
	unsigned char blk_crc[2];
	long long nbit_total;           /* bits produced so far (not in the reference; bookkeeping) */
	long long nsamp_total;          /* 12.5 kHz samples consumed by earlier orc_demod_msk calls (bookkeeping) */
	long long cur_sample;           /* index of the sample being processed (bookkeeping) */
	long long soh_sample;           /* cur_sample when the running block's SOH byte completed (acars.c:290) */

	/* sinks (may be NULL) */
	orc_bit *bitlog; size_t bitlog_cap, bitlog_n;
	orc_frame *frames; size_t frames_cap, frames_n;
} orc_chan;

/* msk.c:30-51 + acars.c:218-237 */
void orc_chan_init(orc_chan *ch, int chn);
/* msk.c:44-48: the matched-filter prototype h[133] */
void orc_msk_h(float *h);
/* msk.c:67-137 (+ putbit msk.c:53-63, decodeAcars acars.c:246-375) */
void orc_demod_msk(orc_chan *ch, const float *dm, int len);

/* rtl.c:283-286: taps for one channel. wf is [M][2] (re,im) */
void orc_rtl_taps(int Fr, int Fc, int M, float *wf);
/* rtl.c:131-168. Fd is sorted in place. rtlInRate = INTRATE*M */
int orc_choose_fc(unsigned int *Fd, unsigned int nbch, int rtlInRate);
/* rtl.c:332-354 for one channel: nout windows of M samples, first ntaps (<=M) used. */
void orc_fir_u8(const uint8_t *iq, size_t nout, int M, int ntaps,
		const float *wf, float *dm);
/* the whole in_callback for nch channels sharing one stream (rtl.c:314-361) */
void orc_in_callback(orc_chan *chs, int nch, const uint8_t *iq, int nout, int M,
		     const float *wf /* [nch][M][2] */, float *dm_scratch /* [nch][nout] */);

/* soapy.c:163-166 oscillator table; soapy.c:232-254 CS16 down-converter (whole stream, any read size) */
void orc_soapy_taps(float Fr, int freq, int M, float *osc);
void orc_fir_cs16(const int16_t *iq, size_t nout, int M, const float *osc, float *dm);

/* sdrplay.c:160-164 / 215-236 (fixed rateMult 160) ; air.c:40-63, 278-285, 299-338 */
void orc_sdrplay_taps(float Fr, unsigned int Fc, float *osc);
void orc_fir_split16(const int16_t *xi, const int16_t *xq, size_t nout, int M, const float *osc, float *dm);
unsigned int orc_air_choose_fc(unsigned int minF, unsigned int maxF);
void orc_air_taps(int Fr, int Fc, unsigned int inrate, float *wf);
void orc_fir_f32r(const float *x, size_t nout, int M, const float *wf, float *dm);

/* acars.c:123-207 parity + CRC verdict of a queued block (no repair):
 * returns 0 if it would be output with err==0, >0 = number of parity errors,
 * -1 = dropped (too short), -2 = crc error with clean parity. */
int orc_frame_check(const orc_frame *f);
unsigned short orc_crc_update(unsigned short crc, unsigned char c);   /* syndrom.h:49 */
/* syndrom.h:52-295 regenerated from its definition (n <= 1936 entries) */
void orc_syndrome_table(unsigned short *out, int n);
/* acars.c:123-207 blk_thread body: 1 + *out = the block outputmsg() gets, 0 = dropped */
int orc_blk_process(const orc_frame *in, orc_frame *out);

#ifdef __cplusplus
}
#endif
/* output.c:486-560,566-568,623-631 (the build without libacars): the fields outputmsg() splits a processed
 * block into -- what every sink (printmsg, buildjson, Netout*) formats afterwards. */
typedef struct {
	int chn, err;
	float lvl;
	int txt_len;
	char mode;
	char addr[8];
	char ack;
	char label[3];
	char bid;
	char no[5];
	char fid[7];
	char bs, be;
	char down;
	char txt[256];
} orc_msg;
void orc_msg_split(const orc_frame *blk, orc_msg *out);

#endif
