/*
 * Translation unit that pulls the reference's rtl.c in unmodified (from -I<reference>)
 * so that its static in_callback() can be reached.  TEST INFRASTRUCTURE ONLY.
 */
#include "rtl.c"

/* rtl.c:314 in_callback is static: export a trampoline */
void ref_rtl_in_callback(unsigned char *buf, uint32_t nread)
{
	in_callback(buf, nread, NULL);
}

int ref_rtl_inrate(void) { return rtlInRate; }
int ref_rtl_inbufsize(void) { return rtlInBufSize; }
