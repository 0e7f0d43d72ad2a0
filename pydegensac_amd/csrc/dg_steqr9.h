/* dsteqr('V') for n = 9, written around its one long dependency chain.
 *
 * LAPACK's implicit QL/QR iteration (netlib 3.12 dsteqr, as called inside dsyev by lap_eig, degensac/lapwrap.c:67-96) is
 * a scalar recurrence: rotation i needs the (c, s, g, p) of rotation i+-1, about 25 dependent fp64 operations each.
 * Everything else is kept off that chain:
 *   - d[] and e[] live in memory every lane can address with a run-time index (LDS on the device; all lanes of the
 *     wave run the recurrence redundantly and store identical values), so the loops stay rolled and the code small;
 *   - the operands of rotation i-+1 (one subdiagonal and one diagonal entry, and this lane's entry of the next
 *     eigenvector column) are loaded while rotation i is still in flight; the third operand of a step is the
 *     previous step's diagonal entry, carried in a register;
 *   - the plane rotations of the eigenvector matrix — lane r owns row r of Z — are applied inside the same step:
 *     consecutive rotations share one column, which is carried in a register instead of being stored and re-read.
 * Same floating-point operations in the same order as the reference, per output number.
 *
 * Host-compilable: tests/test_steqr_cpu.py runs it against the CPU oracle's dsteqr bit for bit.  The includer may
 * override (defaults = plain scalar code)
 *   DG_STEQR_FN                      function qualifiers
 *   DG_STEQR_PTR                     the pointer type of d / e / z (an LDS-qualified pointer on the device)
 *   DG_STEQR_LARTG(f, g, c, s, r)    dlartg (the device passes its reciprocal-sharing form)
 *   DG_STEQR_ST / DG_STEQR_STZ       stores to d / e and to this lane's row of Z
 *   DG_STEQR_ANY(cond)               a wave-uniform condition as a scalar (ballot on the device)
 *   DG_STEQR_FIND_SPLIT / _QL / _QR  the three searches for a negligible subdiagonal entry (one candidate per lane and
 *                                    a ballot on the device)
 * and provides dg_sign / dg_lapy2 / dg_lartg / dg_laev2 / DG_EPS / DG_SAFMIN.
 */
#ifndef DG_STEQR9_H
#define DG_STEQR9_H

#ifndef DG_STEQR_T
#define DG_STEQR_T(i)
#endif
#ifndef DG_STEQR_FN
#define DG_STEQR_FN static inline
#endif
#ifndef DG_STEQR_PTR
#define DG_STEQR_PTR double *
#endif
#ifndef DG_STEQR_LARTG
#define DG_STEQR_LARTG(f, g, c, s, r) dg_lartg((f), (g), (c), (s), (r))
#endif
#ifndef DG_STEQR_ST
/* stores to d / e and to this lane's row of Z.  On the device every lane stores (the same value to the same address):
 * restricting them to one / nine lanes was measured 8 % slower (exec-mask round trips inside the chain) */
#define DG_STEQR_ST(lhs, v) ((lhs) = (v))
#define DG_STEQR_STZ(lhs, v) ((lhs) = (v))
#endif
#ifndef DG_STEQR_ANY
#define DG_STEQR_ANY(cond) (cond)
#endif
#ifndef DG_STEQR_FIND_SPLIT
/* first i in [l1, 8) whose subdiagonal entry is negligible against its two diagonal neighbours, else 8 */
#define DG_STEQR_FIND_SPLIT(d, e, l1, m) do { (m) = 8; \
        for (int i_ = (l1); i_ < 8; i_++) { const double ae_ = fabs((e)[i_]); \
            if (ae_ == 0. || ae_ <= (sqrt(fabs((d)[i_])) * sqrt(fabs((d)[i_+1]))) * DG_EPS) { (m) = i_; break; } } } while (0)
/* QL: first i in [l, lend) with e_i^2 <= (eps^2 |d_i|) |d_i+1| + safmin, else lend */
#define DG_STEQR_FIND_QL(d, e, l, lend, m) do { (m) = (lend); \
        for (int i_ = (l); i_ < (lend); i_++) { double t2_ = fabs((e)[i_]); t2_ *= t2_; \
            if (t2_ <= ((DG_EPS*DG_EPS) * fabs((d)[i_])) * fabs((d)[i_+1]) + DG_SAFMIN) { (m) = i_; break; } } } while (0)
/* QR: last i in [lend, l) with e_i^2 <= (eps^2 |d_i+1|) |d_i| + safmin gives m = i + 1, else lend */
#define DG_STEQR_FIND_QR(d, e, l, lend, m) do { (m) = (lend); \
        for (int i_ = (l) - 1; i_ >= (lend); i_--) { double t2_ = fabs((e)[i_]); t2_ *= t2_; \
            if (t2_ <= ((DG_EPS*DG_EPS) * fabs((d)[i_+1])) * fabs((d)[i_]) + DG_SAFMIN) { (m) = i_ + 1; break; } } } while (0)
#endif

/* d[9], e[>=8] in/out; z[j*zs], j = 0..8: this lane's row of the accumulated orthogonal matrix (in/out).
 * Returns 1 if the iteration limit was hit (dsteqr's info > 0), else 0.  The final ordering is the caller's. */
DG_STEQR_FN int dg_steqr9(DG_STEQR_PTR d, DG_STEQR_PTR e, DG_STEQR_PTR z, const int zs, const int lane)
{
    const int n = 9, nmaxit = n * 30;
    int jtot = 0, l1 = 0, l, m, lsv, lend, lendsv, i;
    double p, g, r, c, s, f, b, rt1, rt2;
    (void)lane;
    while (l1 < n) {
        if (l1 > 0) DG_STEQR_ST(e[l1 - 1], 0.);
        DG_STEQR_FIND_SPLIT(d, e, l1, m);
        if (m < n - 1) DG_STEQR_ST(e[m], 0.);
        l = l1; lsv = l; lend = m; lendsv = lend; l1 = m + 1;
        if (lend == l) continue;
        if (DG_STEQR_ANY(fabs(d[lend]) < fabs(d[l]))) { lend = lsv; l = lendsv; }
        if (lend > l) {
            /* ---- QL: chase the bulge from m-1 down to l ---- */
            for (;;) {
                if (l != lend) DG_STEQR_FIND_QL(d, e, l, lend, m); else m = lend;
                if (m < lend) DG_STEQR_ST(e[m], 0.);
                p = d[l];
                if (m == l) { l++; if (l <= lend) continue; break; }
                if (m == l + 1) {
                    dg_laev2(d[l], e[l], d[l+1], &rt1, &rt2, &c, &s);
                    if (c != 1. || s != 0.) { const double t = z[(l+1)*zs], u = z[l*zs]; DG_STEQR_STZ(z[(l+1)*zs], c*t - s*u); DG_STEQR_STZ(z[l*zs], s*t + c*u);
                        }
                    DG_STEQR_ST(d[l], rt1); DG_STEQR_ST(d[l+1], rt2); DG_STEQR_ST(e[l], 0.);
                    l += 2; if (l <= lend) continue; break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                { const double el = e[l];
                  g = (d[l+1] - p) / (2. * el);
                  r = dg_lapy2(g, 1.);
                  g = d[m] - p + (el / (g + dg_sign(r, g))); }
                s = 1.; c = 1.; p = 0.;
                DG_STEQR_T(0);
                double ei = e[m-1], di = d[m-1], di1 = d[m], zhi = z[m*zs], zlo = z[(m-1)*zs];
_Pragma("unroll 2")      /* two rotations per trip: half the loop-carried register copies (chains 28.9 -> 26.5 us per solve) */
                for (i = m - 1; i >= l; i--) {
                    double ei_n = 0., di_n = 0., zlo_n = 0.;
                    if (i > l) { ei_n = e[i-1]; di_n = d[i-1]; zlo_n = z[(i-1)*zs]; }
                    f = s * ei; b = c * ei;
                    DG_STEQR_LARTG(g, f, &c, &s, &r);
                    if (i != m - 1) DG_STEQR_ST(e[i+1], r);
                    g = di1 - p;
                    r = (di - g)*s + 2.*c*b;
                    p = s * r;
                    DG_STEQR_ST(d[i+1], g + p);
                    g = c*r - b;
                    {   /* columns (i, i+1) with (c, -s); the new column i is the next step's column i+1 */
                        const double ct = c, st = -s; double nhi = zhi, carry = zlo;
                        if (ct != 1. || st != 0.) { nhi = ct*zhi - st*zlo; carry = st*zhi + ct*zlo; }
                        DG_STEQR_STZ(z[(i+1)*zs], nhi); zhi = carry;
                    }
                    ei = ei_n; di1 = di; di = di_n; zlo = zlo_n;
                }
                DG_STEQR_STZ(z[l*zs], zhi);
                DG_STEQR_ST(d[l], di1 - p); DG_STEQR_ST(e[l], g);
                DG_STEQR_T(1);
            }
        } else {
            /* ---- QR: chase the bulge from m up to l-1 ---- */
            for (;;) {
                if (l != lend) DG_STEQR_FIND_QR(d, e, l, lend, m); else m = lend;
                if (m > lend) DG_STEQR_ST(e[m-1], 0.);
                p = d[l];
                if (m == l) { l--; if (l >= lend) continue; break; }
                if (m == l - 1) {
                    dg_laev2(d[l-1], e[l-1], d[l], &rt1, &rt2, &c, &s);
                    if (c != 1. || s != 0.) { const double t = z[l*zs], u = z[(l-1)*zs]; DG_STEQR_STZ(z[l*zs], c*t - s*u); DG_STEQR_STZ(z[(l-1)*zs], s*t + c*u);
                        }
                    DG_STEQR_ST(d[l-1], rt1); DG_STEQR_ST(d[l], rt2); DG_STEQR_ST(e[l-1], 0.);
                    l -= 2; if (l >= lend) continue; break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                { const double el = e[l-1];
                  g = (d[l-1] - p) / (2. * el);
                  r = dg_lapy2(g, 1.);
                  g = d[m] - p + (el / (g + dg_sign(r, g))); }
                s = 1.; c = 1.; p = 0.;
                DG_STEQR_T(0);
                double ei = e[m], di = d[m], di1 = d[m+1], zlo = z[m*zs], zhi = z[(m+1)*zs];
_Pragma("unroll 2")
                for (i = m; i <= l - 1; i++) {
                    double ei_n = 0., di1_n = 0., zhi_n = 0.;
                    if (i < l - 1) { ei_n = e[i+1]; di1_n = d[i+2]; zhi_n = z[(i+2)*zs]; }
                    f = s * ei; b = c * ei;
                    DG_STEQR_LARTG(g, f, &c, &s, &r);
                    if (i != m) DG_STEQR_ST(e[i-1], r);
                    g = di - p;
                    r = (di1 - g)*s + 2.*c*b;
                    p = s * r;
                    DG_STEQR_ST(d[i], g + p);
                    g = c*r - b;
                    {   /* columns (i, i+1) with (c, s); the new column i+1 is the next step's column i */
                        double nlo = zlo, carry = zhi;
                        if (c != 1. || s != 0.) { carry = c*zhi - s*zlo; nlo = s*zhi + c*zlo; }
                        DG_STEQR_STZ(z[i*zs], nlo); zlo = carry;
                    }
                    ei = ei_n; di = di1; di1 = di1_n; zhi = zhi_n;
                }
                DG_STEQR_STZ(z[l*zs], zlo);
                DG_STEQR_ST(d[l], di - p); DG_STEQR_ST(e[l-1], g);
                DG_STEQR_T(1);
            }
        }
        if (jtot >= nmaxit) break;
    }
    return jtot >= nmaxit ? 1 : 0;
}

#endif /* DG_STEQR9_H */
