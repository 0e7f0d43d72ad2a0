// Host build of pydegensac_amd/csrc/dg_steqr9.h for tests/test_steqr_cpu.py (g++ -O2 -ffp-contract=off).
// Test infrastructure: the device's dsteqr against the CPU oracle's dsyev restatement (oracle/dg_small.h)
// on tridiagonal input, where dsytd2/dorg2l reduce to the identity and dsyev == dsteqr + the final ordering.
#include <math.h>
#include <string.h>
extern "C" {
#include "../oracle/dg_small.h"
}
#include "../pydegensac_amd/csrc/dg_steqr9.h"

extern "C" {
// d[9], e[8] -> oracle: w[9] ascending, V column-major 9x9 (V[j*9+i] = component i of eigenvector j); returns info
int t_oracle_tridiag(const double *d, const double *e, double *w, double *V)
{
    double a[81];
    memset(a, 0, sizeof a);
    for (int i = 0; i < 9; i++) a[i*9 + i] = d[i];
    for (int i = 0; i < 8; i++) { a[(i+1)*9 + i] = e[i]; a[i*9 + (i+1)] = e[i]; }
    int info = dg_eig_sym(a, w, 9);
    memcpy(V, a, sizeof a);
    return info;
}
// the routine under test, row by row as the lanes of a wave would run it, then dsteqr's selection sort
int t_steqr9(const double *d_in, const double *e_in, double *w, double *V)
{
    double Z[81]; int info = 0; double dd[9];
    for (int r = 0; r < 9; r++) {
        double d[9], e[9], z[9];
        for (int i = 0; i < 9; i++) { d[i] = d_in[i]; e[i] = i < 8 ? e_in[i] : 0.; z[i] = i == r ? 1. : 0.; }
        info |= dg_steqr9(d, e, z, 1, 0);
        for (int c = 0; c < 9; c++) Z[c*9 + r] = z[c];
        if (r == 0) memcpy(dd, d, sizeof dd);
        else if (memcmp(dd, d, sizeof dd)) return -99;       // the recurrence must not depend on the row
    }
    for (int ii = 1; ii < 9; ii++) {
        int i = ii - 1, k = i; double p = dd[i];
        for (int j = ii; j < 9; j++) if (dd[j] < p) { k = j; p = dd[j]; }
        if (k != i) { dd[k] = dd[i]; dd[i] = p; for (int j = 0; j < 9; j++) { double t = Z[i*9 + j]; Z[i*9 + j] = Z[k*9 + j]; Z[k*9 + j] = t; } }
    }
    memcpy(w, dd, sizeof dd); memcpy(V, Z, sizeof Z);
    return info;
}
}
