/*
 * Pulls the reference's soapy.c in unmodified so that its static reader loop (the DSP lives inside
 * readThreadEntryPoint, soapy.c:205-261) can be driven from a test.  TEST INFRASTRUCTURE ONLY.
 */
#include "soapy.c"

void ref_soapy_run_reader(void)
{
	readThreadEntryPoint(NULL);      /* loops over SoapySDRDevice_readStream until the stub runs dry */
}
int ref_soapy_inrate(void) { return soapyInRate; }
