/* Stand-in for SoapySDR/Formats.h (oracle build only). */
#ifndef ORACLE_STUB_SOAPY_FORMATS_H
#define ORACLE_STUB_SOAPY_FORMATS_H
#define SOAPY_SDR_CS16 "CS16"
#define SOAPY_SDR_CF32 "CF32"
#endif
