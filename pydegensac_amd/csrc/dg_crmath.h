/* Correctly rounded pow(x, 1.0/3), acos and cos on the ranges rroots3 (Ftools.c:251-298) calls them with.
 *
 * Everything else on the path is + - * / sqrt in the reference's order, so it carries the reference's bits.  These three
 * come from the host's libm there (glibc: within 0.52-0.55 ulp, i.e. the correctly rounded value in all but ~0.1-0.2 % of
 * the calls) and from the device's math library here (within 1-2 ulp: one root in 29 differed in its last bits, and a model
 * that differs in its last bits can, very rarely, answer a knife-edge question of the driver differently: DESIGN.md 4).
 * Rounding the exact value to nearest puts the device on the value glibc returns whenever glibc is right.
 *
 * Method: double-double (two_sum / two_prod with fma) Taylor kernels for sin and cos on |r| <= pi/4 after an exact
 * reduction against pi/2 or pi (the first subtraction is exact by Sterbenz' lemma); acos by one Newton step from the
 * library's value on cos (|c| <= 1/2) or on sin^2 of the half angle (|c| > 1/2, where cos alone loses the angle's relative
 * accuracy); x^(1.0/3) by one Newton step of the cube root from the library's cbrt towards x (1 - 3 eps ln x) with
 * eps = 1/3 - (1.0/3) = 2^-54 / 3.  Error of each kernel about 2^-70 of the result: misrounding needs the exact value within
 * that of a rounding boundary (measured against 200-bit arithmetic: tests/test_crmath_cpu.py, none in 3 x 200 000 arguments).
 * Compiles for the host as well (the CPU test does that); the library functions it starts from need only be good to a few ulp. */
#ifndef DG_CRMATH_H
#define DG_CRMATH_H
#include <math.h>
#ifdef __HIPCC__
#define DG_CR_FN static __device__ __forceinline__
#else
#define DG_CR_FN static inline
#endif

typedef struct { double h, l; } dg_dd;

DG_CR_FN dg_dd dg_dd_fast2sum(double a, double b) { dg_dd r; r.h = a + b; r.l = b - (r.h - a); return r; }       /* |a| >= |b| */
DG_CR_FN dg_dd dg_dd_2sum(double a, double b) { dg_dd r; double bb; r.h = a + b; bb = r.h - a; r.l = (a - (r.h - bb)) + (b - bb); return r; }
DG_CR_FN dg_dd dg_dd_2prod(double a, double b) { dg_dd r; r.h = a * b; r.l = fma(a, b, -r.h); return r; }
DG_CR_FN dg_dd dg_dd_mul(dg_dd a, dg_dd b) { dg_dd p = dg_dd_2prod(a.h, b.h); p.l += a.h * b.l + a.l * b.h; return dg_dd_fast2sum(p.h, p.l); }
DG_CR_FN dg_dd dg_dd_add(dg_dd a, dg_dd b) { dg_dd s = dg_dd_2sum(a.h, b.h); s.l += a.l + b.l; return dg_dd_fast2sum(s.h, s.l); }
DG_CR_FN dg_dd dg_dd_sq(dg_dd r) { dg_dd z = dg_dd_2prod(r.h, r.h); z.l += 2.0 * r.h * r.l; return dg_dd_fast2sum(z.h, z.l); }

/* cos r, |r| <= 0.79:  1 + z (-1/2 + z (1/24 + z (-1/720 + z q(z)))),  z = r^2, q = the series' terms 8 ... 20 in double */
DG_CR_FN dg_dd dg_cr_cos_k(dg_dd r)
{
    const dg_dd z = dg_dd_sq(r);
    double q = 0x1.e542ba4020225p-62;
    q = fma(q, z.h, -0x1.6827863b97d97p-53); q = fma(q, z.h, 0x1.ae7f3e733b81fp-45); q = fma(q, z.h, -0x1.93974a8c07c9dp-37);
    q = fma(q, z.h, 0x1.1eed8eff8d898p-29); q = fma(q, z.h, -0x1.27e4fb7789f5cp-22); q = fma(q, z.h, 0x1.a01a01a01a01ap-16);
    dg_dd a = dg_dd_2sum(-0x1.6c16c16c16c17p-10, z.h * q); a.l += 0x1.f49f49f49f49fp-65; a = dg_dd_fast2sum(a.h, a.l);
    const dg_dd c2 = {0x1.5555555555555p-5, 0x1.5555555555555p-59}, mh = {-0.5, 0.0}, one = {1.0, 0.0};
    a = dg_dd_add(c2, dg_dd_mul(a, z));
    a = dg_dd_add(mh, dg_dd_mul(a, z));
    return dg_dd_add(one, dg_dd_mul(a, z));
}

/* sin r, |r| <= 0.79:  r + r z (-1/6 + z (1/120 + z (-1/5040 + z q(z)))) */
DG_CR_FN dg_dd dg_cr_sin_k(dg_dd r)
{
    const dg_dd z = dg_dd_sq(r);
    double q = 0x1.71b8ef6dcf572p-66;
    q = fma(q, z.h, -0x1.2f49b46814157p-57); q = fma(q, z.h, 0x1.952c77030ad4ap-49); q = fma(q, z.h, -0x1.ae7f3e733b81fp-41);
    q = fma(q, z.h, 0x1.6124613a86d09p-33); q = fma(q, z.h, -0x1.ae64567f544e4p-26); q = fma(q, z.h, 0x1.71de3a556c734p-19);
    dg_dd a = dg_dd_2sum(-0x1.a01a01a01a01ap-13, z.h * q); a.l += -0x1.a01a01a01a01ap-73; a = dg_dd_fast2sum(a.h, a.l);
    const dg_dd s2 = {0x1.1111111111111p-7, 0x1.1111111111111p-63}, s1 = {-0x1.5555555555555p-3, -0x1.5555555555555p-57};
    a = dg_dd_add(s2, dg_dd_mul(a, z));
    a = dg_dd_add(s1, dg_dd_mul(a, z));
    a = dg_dd_mul(a, z);
    return dg_dd_add(r, dg_dd_mul(r, a));
}

#define DG_CR_PIO2_1 0x1.921fb54442d18p+0
#define DG_CR_PIO2_2 0x1.1a62633145c07p-54
#define DG_CR_PIO2_3 (-0x1.f1976b7ed8fbcp-110)

/* cos x as a double-double, 0 <= x <= pi */
DG_CR_FN dg_dd dg_cr_cos_dd(double x)
{
    if (x < 0.5 * DG_CR_PIO2_1) { const dg_dd r = {x, 0.0}; return dg_cr_cos_k(r); }
    if (x <= 0x1.2d97c7f3321d2p+1) {                      /* <= 3 pi / 4:  cos x = sin(pi/2 - x), the first difference exact */
        dg_dd r = dg_dd_2sum(DG_CR_PIO2_1 - x, DG_CR_PIO2_2); r.l += DG_CR_PIO2_3; r = dg_dd_fast2sum(r.h, r.l);
        return dg_cr_sin_k(r);
    }
    dg_dd r = dg_dd_2sum(2.0 * DG_CR_PIO2_1 - x, 2.0 * DG_CR_PIO2_2); r.l += 2.0 * DG_CR_PIO2_3; r = dg_dd_fast2sum(r.h, r.l);
    r = dg_cr_cos_k(r);                                   /* cos x = -cos(pi - x) */
    r.h = -r.h; r.l = -r.l; return r;
}

DG_CR_FN double dg_cr_cos(double x)
{
    x = fabs(x);                                                      /* PIT - phit can be -9e-16 at cosphi = -1 */
    if (!(x <= 2.0 * DG_CR_PIO2_1)) return cos(x);                    /* not an angle rroots3 produces: the library's answer */
    const dg_dd c = dg_cr_cos_dd(x);
    return c.h + c.l;
}

DG_CR_FN double dg_cr_acos(double c)
{
    const double y0 = acos(c);
    if (!(c > -1.0 && c < 1.0)) return y0;                             /* +-1 (exact in every library), out of range, nan */
    if (c >= -0.5 && c <= 0.5) {
        /* Newton on cos phi = c:  phi1 = phi0 + (cos phi0 - c) / sin phi0 */
        const dg_dd cd = dg_cr_cos_dd(y0);
        const double num = (cd.h - c) + cd.l;
        return y0 + num / sqrt((1.0 - c) * (1.0 + c));
    }
    /* half angle: t = (1 - |c|) / 2 = sin^2 psi exactly; Newton on sin^2 psi = t keeps psi's RELATIVE accuracy near c = +-1 */
    const double t = (1.0 - fabs(c)) * 0.5;
    const double p0 = asin(sqrt(t));
    const dg_dd pd = {p0, 0.0};
    const dg_dd s = dg_cr_sin_k(pd), s2 = dg_dd_mul(s, s);
    const double delta = ((s2.h - t) + s2.l) / (2.0 * s.h * sqrt(1.0 - t));       /* psi1 = psi0 - delta */
    if (c > 0) return 2.0 * (p0 - delta);
    dg_dd r = dg_dd_2sum(2.0 * DG_CR_PIO2_1, -2.0 * p0);                          /* pi - 2 psi1 */
    r.l += 2.0 * DG_CR_PIO2_2 + 2.0 * delta;
    return r.h + r.l;
}

/* pow(A, 1.0/3), A > 0:  1.0/3 = 1/3 - eps with 3 eps = 2^-54, so A^(1.0/3) = cbrt(A (1 - 3 eps ln A)) up to 1e-28 */
DG_CR_FN double dg_cr_pow13(double A)
{
    if (!(A > 1e-290 && A < 1e290)) return pow(A, 1.0 / 3);
    const double y0 = cbrt(A);
    const dg_dd p = dg_dd_2prod(y0, y0);
    const double qh = p.h * y0, ql = fma(p.h, y0, -qh) + p.l * y0;                  /* y0^3 = qh + ql */
    const double res = (qh - A) + (ql + A * (0x1p-54 * log(A)));                   /* y0^3 - A (1 - 3 eps ln A) */
    return y0 - res / (3.0 * p.h);
}

#endif /* DG_CRMATH_H */
