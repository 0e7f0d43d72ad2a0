// Host build of pydegensac_amd/csrc/dg_mat3.h for tests/test_mat3_cpu.py (g++ -O2 -ffp-contract=off).
// Test infrastructure: exports plain C wrappers so the routines can be compared with oracle/_ref bit for bit.
#include "../pydegensac_amd/csrc/dg_mat3.h"
extern "C" {
int t_inv3(double *a) { return dg_inv3(a); }
void t_svd3_right(double *a, double *v, double *d) { dg_svd3_right(a, v, d); }
void t_hdetect(const double *F, const double *u7x4, const unsigned char *idxs, double *H) { dg_Hdetect(F, (const double (*)[4])u7x4, idxs, H); }
int t_null9(int rows, double *M, double *ns)
{
    if (rows == 7) return dg_null9<7, 2>(M, ns);
    if (rows == 8) return dg_null9<8, 2>(M, ns);
    return dg_null9<9, 2>(M, ns);
}
}
