/* Pulls the reference's sdrplay.c in unmodified to reach its static stream callback
 * (sdrplay.c:202-237).  TEST INFRASTRUCTURE ONLY. */
#include "sdrplay.c"
void ref_sdrplay_callback(int16_t *xi, int16_t *xq, uint32_t n)
{
	myStreamCallback(xi, xq, 0, 0, 0, 0, n, 0, 0, NULL);
}
unsigned int ref_sdrplay_fc(void) { return Fc; }
/* Overwrites the oscillator table initSdrplay built for channel n (sdrplay.c:160-165) with the caller's, like ref_set_oscillator
 * on the soapy.c path: the reference's callback on the very table another implementation was given.  Harness only. */
int ref_set_oscillator(int n, const float *taps, int ntaps)
{
	int i;
	if (n < 0 || (unsigned int)n >= nbch || ntaps < 0 || ntaps > SDRPLAY_MULT)
		return -1;
	for (i = 0; i < SDRPLAY_MULT; i++)
		channel[n].oscillator[i] = i < ntaps ? taps[2 * i] + taps[2 * i + 1] * I : 0;
	return 0;
}
