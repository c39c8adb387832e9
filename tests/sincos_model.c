// CPU model of the demodulator's device sin/cos (acarsdec_amd/csrc/msk.hip sincos_tab, table from host_setup.c
// acg_host_sincos_table): the same operations in the same order (fma = one rounding).  Test infrastructure: built and run by
// tests/test_host_logic.py; prints the worst error in ulp against long-double libm and how many of the float-rounded mixer
// products (msk.c:90: what the demodulator keeps) differ from glibc's cexp.  usage: sincos_model [samples]
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <complex.h>
static double tab[128][2];
void acg_host_sincos_table(double* t);       // the product's own table builder (acarsdec_amd/csrc/host_setup.c, linked in)
static void build(void) { acg_host_sincos_table(&tab[0][0]); }
static inline void sc_tab(double x, double* sn, double* cs) {
    const double kd = rint(x * (6.36619772367581382433e-01 * 32.0));
    const int q = (int)kd;
    double r = fma(-kd, 1.57079632673412561417e+00 / 32.0, x);
    r = fma(-kd, 6.07710050650619224932e-11 / 32.0, r);
    const double cj = tab[q & 127][0], sj = tab[q & 127][1];
    const double z = r * r;
    // sin r - r = r z (S1 + z (S2 + z S3)),  cos r - 1 = z (C0 + z (C1 + z C2))
    double ps = fma(z, -1.98412698412698412698e-04, 8.33333333333333333333e-03);
    ps = fma(z, ps, -1.66666666666666666667e-01);
    const double sm = (r * z) * ps;            // sin r - r
    double pc = fma(z, -1.38888888888888888889e-03, 4.16666666666666666667e-02);
    pc = fma(z, pc, -0.5);
    const double cm1 = z * pc;                 // cos r - 1
    const double sr = r + sm;
    // cos p = cj + (cj (cos r - 1) - sj sin r);  sin p = sj + (sj (cos r - 1) + cj sin r)
    const double u = fma(cj, cm1, -(sj * sr));
    const double w = fma(sj, cm1, cj * sr);
    *cs = cj + u;
    *sn = sj + w;
}
static double ulp_of(double v) { double a = fabs(v); if (a < 1e-300) return 4.9e-324; int e; frexp(a, &e); return ldexp(1.0, e - 53); }
int main(int argc, char** argv) {
    build();
    uint64_t st = 88172645463325252ull;
    double maxs = 0, maxc = 0; long nid_s = 0, nid_c = 0, n = argc > 1 ? atol(argv[1]) : 20000000, prodmis = 0, prodn = 0;
    for (long i = 0; i < n; i++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double x = (double)(st >> 11) / 9007199254740992.0 * 6.283185307179586;
        double s, c; sc_tab(x, &s, &c);
        const long double ls = sinl((long double)x), lc = cosl((long double)x);
        const double es = fabs((double)((long double)s - ls)) / ulp_of((double)ls), ec = fabs((double)((long double)c - lc)) / ulp_of((double)lc);
        if (fabs((double)ls) > 1e-3 && es > maxs) maxs = es;
        if (fabs((double)lc) > 1e-3 && ec > maxc) maxc = ec;
        nid_s += s == sin(x); nid_c += c == cos(x);
        // what the demodulator keeps: (float)(in * cos), (float)(in * -sin) against glibc cexp (msk.c:90)
        const float in = (float)(0.001 + (double)((st >> 20) & 0xffff) / 65536.0);
        const double complex e = cexp(-x * I);
        const float r0 = (float)((double)in * creal(e)), i0 = (float)((double)in * cimag(e));
        const float r1 = (float)((double)in * c), i1 = (float)((double)in * (-s));
        prodmis += (r0 != r1) + (i0 != i1); prodn += 2;
    }
    printf("max error (|value| > 1e-3): sin %.3f ulp, cos %.3f ulp; bit-identical to libm: sin %.1f %%, cos %.1f %%\n", maxs, maxc, 100.0 * nid_s / n, 100.0 * nid_c / n);
    printf("float-rounded mixer products that differ from glibc cexp: %ld of %ld\n", prodmis, prodn);
    // edges
    const double xs[] = {0.0, 0x1p-60, 3.14159265358979323846 / 256, 3.14159265358979323846, 6.283185307179586 - 1e-15, 6.283185307179586, 1.5707963267948966, 4.71238898038469};
    for (int k = 0; k < 8; k++) { double s, c; sc_tab(xs[k], &s, &c); printf("x %.17g: sin %.17g (libm %.17g) cos %.17g (libm %.17g)\n", xs[k], s, sin(xs[k]), c, cos(xs[k])); }
    return 0;
}
