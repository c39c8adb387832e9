/*
 * Stand-in for SoapySDR/Device.h, written for the oracle build only.  Declares the entry points
 * the reference's soapy.c calls; ref_glue.c defines them: set-up calls do nothing, readStream
 * serves samples from a buffer supplied by the test.  TEST INFRASTRUCTURE ONLY.
 */
#ifndef ORACLE_STUB_SOAPY_DEVICE_H
#define ORACLE_STUB_SOAPY_DEVICE_H
#include <stddef.h>
#include <stdbool.h>
#include "Types.h"
#define SOAPY_SDR_TX 0
#define SOAPY_SDR_RX 1
typedef struct SoapySDRDevice SoapySDRDevice;
typedef struct SoapySDRStream SoapySDRStream;
const char *SoapySDRDevice_lastError(void);
SoapySDRDevice *SoapySDRDevice_makeStrArgs(const char *args);
int SoapySDRDevice_unmake(SoapySDRDevice *device);
int SoapySDRDevice_setGainMode(SoapySDRDevice *device, int direction, size_t channel, bool automatic);
int SoapySDRDevice_setGain(SoapySDRDevice *device, int direction, size_t channel, double value);
int SoapySDRDevice_setFrequencyCorrection(SoapySDRDevice *device, int direction, size_t channel, double value);
int SoapySDRDevice_setFrequency(SoapySDRDevice *device, int direction, size_t channel, double frequency, const SoapySDRKwargs *args);
int SoapySDRDevice_setSampleRate(SoapySDRDevice *device, int direction, size_t channel, double rate);
int SoapySDRDevice_setAntenna(SoapySDRDevice *device, int direction, size_t channel, const char *name);
SoapySDRStream *SoapySDRDevice_setupStream(SoapySDRDevice *device, int direction, const char *format,
					   const size_t *channels, size_t numChans, const SoapySDRKwargs *args);
int SoapySDRDevice_closeStream(SoapySDRDevice *device, SoapySDRStream *stream);
int SoapySDRDevice_activateStream(SoapySDRDevice *device, SoapySDRStream *stream, int flags, long long timeNs, size_t numElems);
int SoapySDRDevice_deactivateStream(SoapySDRDevice *device, SoapySDRStream *stream, int flags, long long timeNs);
int SoapySDRDevice_readStream(SoapySDRDevice *device, SoapySDRStream *stream, void *const *buffs, size_t numElems,
			      int *flags, long long *timeNs, long timeoutUs);
#endif
