/* Pulls the reference's sdrplay.c in unmodified to reach its static stream callback
 * (sdrplay.c:202-237).  TEST INFRASTRUCTURE ONLY. */
#include "sdrplay.c"
void ref_sdrplay_callback(int16_t *xi, int16_t *xq, uint32_t n)
{
	myStreamCallback(xi, xq, 0, 0, 0, 0, n, 0, 0, NULL);
}
unsigned int ref_sdrplay_fc(void) { return Fc; }
