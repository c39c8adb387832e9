/*
 * Stand-in for libairspy's airspy.h (oracle build only).  Declares what the reference's air.c uses;
 * definitions in ref_glue.c do nothing but report one device with one sample rate.
 * TEST INFRASTRUCTURE ONLY.
 */
#ifndef ORACLE_STUB_AIRSPY_H
#define ORACLE_STUB_AIRSPY_H
#include <stdint.h>
#define AIRSPY_SUCCESS 0
#define AIRSPY_TRUE 1
enum airspy_sample_type { AIRSPY_SAMPLE_FLOAT32_IQ = 0, AIRSPY_SAMPLE_FLOAT32_REAL = 1 };
struct airspy_device;
typedef struct {
	struct airspy_device *device;
	void *ctx;
	void *samples;
	int sample_count;
	uint64_t dropped_samples;
	enum airspy_sample_type sample_type;
} airspy_transfer_t, airspy_transfer;
typedef int (*airspy_sample_block_cb_fn)(airspy_transfer *transfer);
int airspy_list_devices(uint64_t *serials, int count);
int airspy_open_sn(struct airspy_device **device, uint64_t serial_number);
int airspy_open(struct airspy_device **device);
int airspy_close(struct airspy_device *device);
int airspy_exit(void);
const char *airspy_error_name(int errcode);
int airspy_set_sample_type(struct airspy_device *device, enum airspy_sample_type sample_type);
int airspy_get_samplerates(struct airspy_device *device, uint32_t *buffer, const uint32_t len);
int airspy_set_samplerate(struct airspy_device *device, uint32_t samplerate);
int airspy_set_packing(struct airspy_device *device, uint8_t value);
int airspy_set_linearity_gain(struct airspy_device *device, uint8_t value);
int airspy_set_vga_gain(struct airspy_device *device, uint8_t value);
int airspy_set_freq(struct airspy_device *device, const uint32_t freq_hz);
int airspy_r820t_write(struct airspy_device *device, uint8_t register_number, uint8_t value);
int airspy_start_rx(struct airspy_device *device, airspy_sample_block_cb_fn callback, void *rx_ctx);
int airspy_is_streaming(struct airspy_device *device);
#endif
