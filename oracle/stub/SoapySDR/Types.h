/* Stand-in for SoapySDR/Types.h (oracle build only; SoapySDR is not installed here). */
#ifndef ORACLE_STUB_SOAPY_TYPES_H
#define ORACLE_STUB_SOAPY_TYPES_H
#include <stddef.h>
typedef struct { size_t size; char **keys; char **vals; } SoapySDRKwargs;
#endif
