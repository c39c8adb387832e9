// acg_internal.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
// Not installed; the public surface is include/acarsdec_amd.h.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define ACG_WG_FIR 256          // FIR workgroup: 4 waves, one 64-window tile, taps split over waves
#define ACG_TILE_WIN 64         // windows (12.5 kHz outputs) per FIR tile = one per lane
#define ACG_MSG_TXT 242         // text bytes of a split message (== ACG_MSGTXTMAX of the public header)
#define ACG_WG_MSK 64           // one MSK wave: 64/LPC channels (LPC lanes per channel); 1 or 4 waves per workgroup

// Per-channel demodulator + framing state, resident in HBM across calls.
// Mirrors the MSK/ACARS fields of channel_t (acarsdec.h:76-89) plus bookkeeping.
struct AcgChan {
    double phi;                 // MskPhi
    double df;                  // MskDf
    double lvlsum;              // MskLvlSum
    float clk;                  // MskClk
    int bitcount;               // MskBitCount
    unsigned int S;             // MskS
    unsigned int idx;           // idx
    float inb[22];              // inb[11] complex
    int nbits;                  // nbits
    int astate;                 // Acarsstate
    int blen;                   // blk->len
    int berr;                   // blk->err
    unsigned int outbits;       // outbits
    unsigned int crc0;          // blk->crc[0] (held between CRC1 and CRC2)
    long long nbit_total;       // bits produced since reset
    long long nsamp_total;      // 12.5 kHz samples consumed since reset
    unsigned int soh32;         // low 32 bits of the sample index at which the block's SOH byte completed (acars.c:290 stamps blk->tv there)
    unsigned int pad_;
};

// One queued block (device layout; converted to acg_frame on the host).
struct AcgFrameRec {
    int chn;
    int len;
    int err;
    int bitcount;               // MskBitCount at acars.c:351
    double lvlsum;              // MskLvlSum at acars.c:351 (host takes 10*log10(lvlsum/bitcount))
    long long end_bit;
    long long end_sample;
    unsigned char crc[2];
    unsigned char status;       // 0 raw (as queued by decodeAcars), 1 processed + kept, 2 processed + dropped
    unsigned char pad[1];
    int soh_back;               // end_sample - (sample at which the SOH byte completed): where the reference stamps blk->tv (acars.c:290)
    unsigned char txt[256];     // 16-byte aligned, 16-byte multiples for vector copies
};

// One split message on the device: the public acg_msg (include/acarsdec_amd.h) field for field, with the two
// operands of the level in place of the padding the host overwrites.
struct AcgMsgRec {
    int chn;
    int err;
    float lvl;                  // filled in on the host: 10*log10(lvlsum/bitcount), acars.c:351
    int txt_len;
    long long end_bit;
    long long end_sample;
    double lvlsum;              // (acg_msg: reserved)
    int bitcount;               // (acg_msg: reserved)
    char valid;                 // (acg_msg: reserved) 0 = dropped by the block repair
    char mode;
    char addr[8];
    char ack;
    char label[3];
    char bid;
    char no[5];
    char fid[7];
    char bs, be;
    char down;
    char txt[ACG_MSG_TXT];
    int soh_back;               // (acg_msg: the host turns it into soh_sample) end_sample - sample of the SOH byte
};

struct FirArgs {
    const uint8_t* iq;          // [nstreams] rows
    size_t pitch;               // bytes between stream rows (multiple of 16)
    const int* stream_of;       // [nch]
    const float* taps;          // [nch][ntaps_pad][2]
    float* dm;                  // [nch][dm_pitch]
    size_t dm_pitch;            // floats
    int nch;
    int decim;                  // M
    int ntaps_pad;              // taps per channel, zero padded to a multiple of 8
    int ntaps;                  // taps per channel that carry values (the exact-order kernel stops there)
    int nwin;                   // outputs per channel this launch
    int nseg;                   // time segments per channel (grid = nch * nseg)
    int row_bytes;              // 2*M
    int row_stride;             // LDS row stride in bytes ((row_stride/16) odd)
    int cpr;                    // 16-byte chunks per row = row_bytes/16
    unsigned int cpr_magic;     // ceil(2^20 / cpr): c / cpr == (c * magic) >> 20 for c < 2^11
    size_t plane;               // FMT_SPLIT: byte distance from a stream's I plane to its Q plane
    const int4* groups;         // shared-stream kernel: {stream, first index into group_ch, channel count (<= 8), 0}
    const int* group_ch;        // shared-stream kernel: channel ids ordered by stream
    int ngroups;
    const float* gtaps;         // shared-stream kernel: taps regrouped [group][step][channel][16]
    int wg_per_cu;              // resident down-converter workgroups per CU (0 = as many as fit, at most 5)
    int ncu;                    // CUs the launch may use (0 = the whole device; fewer on a CU-masked stream)
    int kseg;                   // format kernels: column segments per window (long windows pass through LDS in kseg slices)
    int cpr_total;              // format kernels: 16-byte chunks per whole window (cpr = chunks per slice)
    float out_scale;            // power-of-two scale applied to |D| (1, 1/32768 soapy.c:241, 1/4 sdrplay.c:225)
    unsigned int* work_counter; // run dispenser of the dynamically scheduled kernels (ACG_DISP_WORDS words per launch in flight)
    int stream_identity;        // stream_of[ch] == ch for every channel: the row base needs no lookup
    int high_prio;              // wave-private kernel: raise the wave priority (the demodulator shares its CUs and has slack)
    int run_pairs;              // wave-private kernel: two-tile bodies per dispensed run (set by the launcher)
    int shares_cus;             // demodulator workgroups run on the same CUs (no CU partition): leave them LDS
    const void* mm_img;         // matrix-pipe shared-stream kernel (fir_mm.hip): the groups' A-operand images (tap digits)
    const void* mm_chan;        // ... and MmChan[nch]
};

// fir_mm.hip: what a channel's tap table contributes besides its digits
struct MmChan {
    double scale;               // 2^(e - 30): value of one unit of the 31-bit fixed-point taps
    double dc_re, dc_im;        // (128 - 127.37f) (1 + j) sum w
    double up;                  // 2^(30 - e), 0 for an all-zero table
};

// Run dispenser of one launch in flight: words [0], [1] = {tickets, finished} of the workgroup-granular
// kernels; the wave-granular kernel uses ACG_DISP_SHARDS ticket words, one 128-byte line apart (so that the
// atomics of different shards go to different L2 channels), and a `finished` word after them.
#define ACG_SINCOS_N 128
#define ACG_DISP_SHARDS 8
#define ACG_DISP_STRIDE 32
#define ACG_DISP_WORDS ((ACG_DISP_SHARDS + 1) * ACG_DISP_STRIDE)

struct MskArgs {
    AcgChan* st;                // [nch]
    const float* dm;            // [nch][dm_pitch]
    size_t dm_pitch;
    const float* h;             // [133] matched filter prototype (msk.c:44-48)
    const double* sctab;        // [ACG_SINCOS_N][2] (cos, sin) of j * 2 pi / N (acg_host_sincos_table)
    unsigned char* txt;         // [nch][256] blk->txt being assembled
    AcgFrameRec* frames;        // queue
    unsigned int* frame_count;  // queue length (atomic)
    unsigned int frame_cap;
    float2* bits;               // [nch][bit_cap] {vo, lvl}, may be null
    int* nbits_out;             // [nch]
    int bit_cap;
    int nch;
    int len;                    // samples per channel this launch
    int bit_append;             // 0: bit records start at 0; 1: append after nbits_out[ch] (same call)
    int waves_per_group;        // 4 on a CU-masked stream (placement), else 1
    unsigned int* snap;         // host-mapped word: the last workgroup out publishes the queue length there
    unsigned int* done_ctr;     // workgroups finished (re-armed by the last one)
    int high_prio;              // raise wave priority (latency mode)
    int dm_vec_ok;              // dm rows are 16-byte aligned: the window refill may use float4 loads
    unsigned long long* stamp;  // measurement build (ACG_MSK_STAMP) only: [waves][10] phase cycle sums, else null
    int precise_mixer;          // ACG_F_PRECISE_MIXER: the < 1 ulp polynomial sin/cos instead of table + rotation
};

#ifdef __cplusplus
extern "C" {
#endif
// kernel launchers (fir.hip / msk.hip / synth.hip); stream is a hipStream_t
int acg_launch_fir(const FirArgs* a, void* stream);
int acg_launch_fir_generic(const FirArgs* a, void* stream);
int acg_launch_fir_shared(const FirArgs* a, void* stream);
int acg_launch_regroup_taps(const FirArgs* a, void* stream);          // channels grouped by stream (a->groups)
size_t acg_fir_mm_image_bytes(int decim, int ngroups);               // fir_mm.hip; 0: the matrix kernel does not take this decimation
int acg_fir_mm_takes(const FirArgs* a);
int acg_launch_fir_mm_prep(const FirArgs* a, void* stream);
int acg_launch_fir_mm(const FirArgs* a, void* stream);
size_t acg_fir_mm1_image_bytes(int decim, int nch);                  // fir_mm.hip, one stream per channel: 256 B per channel and k-step
int acg_fir_mm1_takes(const FirArgs* a);
int acg_launch_fir_mm1_prep(const FirArgs* a, void* stream);
int acg_launch_fir_mm1(const FirArgs* a, void* stream);
int acg_launch_fir_fmt(const FirArgs* a, int fmt, void* stream);     // fmt: 1 CS16, 2 split int16 planes, 3 real f32
size_t acg_fir_lds_bytes(const FirArgs* a);
int acg_launch_msk(const MskArgs* a, int lanes_per_channel, void* stream);
int acg_launch_msk_lean(const MskArgs* a, int lanes_per_channel, int waves_per_group, unsigned int grid, void* stream);   // msk_lean.hip: launches without a bit log
int acg_launch_msk2(const MskArgs* a, int pairs_per_group, void* stream);     // msk2.hip: the stream split over two waves (8 lanes per channel)
int acg_launch_blk_repair(AcgFrameRec* frames, unsigned int cap, const unsigned int* upto, unsigned int* done_upto,
                          unsigned int* done_ctr, const unsigned short* synd, const unsigned short* crctab, int nch, void* stream);
int acg_launch_msg_split(const AcgFrameRec* frames, unsigned int cap, unsigned int first, unsigned int n, AcgMsgRec* out, void* stream);
int acg_launch_sincos_selftest(const double* x, double* s, double* c, int n, const double* sctab, void* stream);
int acg_launch_div2_selftest(const double* n0, const double* n1, const double* d, double* out, int n, void* stream);
int acg_launch_synth_iq(uint8_t* iq, size_t pitch, int nrows, int nout, int decim, const float* env,
                        size_t env_pitch, const int* env_index, const float* off_hz, const float* phase,
                        float scale, float noise, uint64_t seed, void* stream);
int acg_launch_fill_random(uint8_t* dev, size_t pitch, int nrows, size_t row_bytes, uint64_t seed, void* stream);
int acg_launch_read_probe(const void* dev, size_t bytes, unsigned int* sink, int ncu, void* stream);
#ifdef __cplusplus
}
#endif
