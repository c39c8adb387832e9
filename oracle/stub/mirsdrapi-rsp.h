/*
 * Stand-in for SDRplay's mirsdrapi-rsp.h (oracle build only; the vendor API is not installed).
 * Declares what the reference's sdrplay.c uses; definitions in ref_glue.c do nothing.
 * TEST INFRASTRUCTURE ONLY.
 */
#ifndef ORACLE_STUB_MIRSDRAPI_RSP_H
#define ORACLE_STUB_MIRSDRAPI_RSP_H
#include <stdint.h>
#define MIR_SDR_API_VERSION 2.13f
typedef enum { mir_sdr_Success = 0, mir_sdr_Fail = 1 } mir_sdr_ErrT;
typedef enum { mir_sdr_BW_1_536 = 1536 } mir_sdr_Bw_MHzT;
typedef enum { mir_sdr_IF_Zero = 0 } mir_sdr_If_kHzT;
typedef enum { mir_sdr_USE_SET_GR = 0, mir_sdr_USE_SET_GR_ALT_MODE = 1, mir_sdr_USE_RSP_SET_GR = 2 } mir_sdr_SetGrModeT;
typedef enum { mir_sdr_AGC_DISABLE = 0, mir_sdr_AGC_100HZ = 1 } mir_sdr_AgcControlT;
typedef struct { char *SerNo; char *DevNm; unsigned char hwVer; unsigned char devAvail; } mir_sdr_DeviceT;
typedef void (*mir_sdr_StreamCallback_t)(short *xi, short *xq, unsigned int firstSampleNum, int grChanged, int rfChanged,
					 int fsChanged, unsigned int numSamples, unsigned int reset, unsigned int hwRemoved, void *cbContext);
typedef void (*mir_sdr_GainChangeCallback_t)(unsigned int gRdB, unsigned int lnaGRdB, void *cbContext);
mir_sdr_ErrT mir_sdr_ApiVersion(float *version);
mir_sdr_ErrT mir_sdr_GetDevices(mir_sdr_DeviceT *devices, unsigned int *numDevs, unsigned int maxDevs);
mir_sdr_ErrT mir_sdr_SetDeviceIdx(unsigned int idx);
mir_sdr_ErrT mir_sdr_ReleaseDeviceIdx(void);
mir_sdr_ErrT mir_sdr_StreamInit(int *gRdB, double fsMHz, double rfMHz, mir_sdr_Bw_MHzT bwType, mir_sdr_If_kHzT ifType, int LNAstate,
				int *gRdBsystem, mir_sdr_SetGrModeT setGrMode, int *samplesPerPacket, mir_sdr_StreamCallback_t StreamCbFn,
				mir_sdr_GainChangeCallback_t GainChangeCbFn, void *cbContext);
mir_sdr_ErrT mir_sdr_AgcControl(mir_sdr_AgcControlT enable, int setPoint_dBfs, int knee_dBfs, unsigned int decay_ms,
				unsigned int hang_ms, int syncUpdate, int LNAstate);
mir_sdr_ErrT mir_sdr_SetPpm(double ppm);
mir_sdr_ErrT mir_sdr_SetDcMode(int dcCal, int speedUp);
mir_sdr_ErrT mir_sdr_SetDcTrackTime(int trackTime);
mir_sdr_ErrT mir_sdr_DCoffsetIQimbalanceControl(unsigned int DCenable, unsigned int IQenable);
#endif
