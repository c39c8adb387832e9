/* Pulls the reference's air.c in unmodified to reach its static receive callback
 * (air.c:291-341).  TEST INFRASTRUCTURE ONLY. */
#include "air.c"
int ref_air_callback(float *samples, int count)
{
	airspy_transfer_t t;
	memset(&t, 0, sizeof(t));
	t.samples = samples;
	t.sample_count = count;
	return rx_callback(&t);
}
unsigned int ref_air_mult(void) { return AIRMULT; }
unsigned int ref_air_inrate(void) { return AIRINRATE; }
