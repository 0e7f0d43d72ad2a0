/* TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference LO-RANSAC / DEGENSAC hot path
 * (see dg_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this. */
#ifndef DG_ORACLE_H
#define DG_ORACLE_H
#include <stdint.h>

typedef struct {
    unsigned I;      /* inliers (rectangular gain)                         rtools.h:18-29 */
    double   J;      /* MSAC truncated-quadratic gain                                      */
    unsigned Is;     /* inliers also consistent under the symmetric check                  */
    unsigned Ilafs;  /* LAF-consistent inliers                                             */
} dg_score;

/* stats[] layout shared with the product's C-ABI (include/mi_degensac.h) */
enum { DG_ST_SAMPLES = 0, DG_ST_LO_RUNS, DG_ST_REJECTED, DG_ST_I, DG_ST_MODELS, DG_ST_DEGEN,
       DG_ST_IH, DG_ST_BEST_SAMPLE, DG_ST_COUNT = 16 };

#ifdef __cplusplus
extern "C" {
#endif

int dg_oracle_find_fundamental(const double *x1, const double *x2, int n, int dim,
                               double px_th, double conf, int max_iters, int error_type,
                               int sym_check, double laf_coef, int degen, unsigned seed,
                               int final_laf_filter,
                               double *F, unsigned char *mask, int *stats);

int dg_oracle_find_homography(const double *x1, const double *x2, int n, int dim,
                              double px_th, double conf, int max_iters, int error_type,
                              int sym_check, double laf_coef, unsigned seed,
                              double *H, unsigned char *mask, int *stats);

/* ranH2el.c:19 ransacH2el (2 ellipse-to-ellipse correspondences per sample); u10 = [n, 10], th on HDs' squared distance */
int dg_oracle_ransacH2el(const double *u10, int n, double th, double conf, int max_iters, int do_lo, int inl_limit, unsigned seed,
                         double *H, unsigned char *mask, int *stats);

/* unit-level entry points used by the tests (thin wrappers over dg_small.h / dg_oracle.c) */
void dg_oracle_rand_stream(unsigned seed, int count, int *out);
int  dg_oracle_sample_stream(unsigned seed, int n, int sample_size, int iters, int *samidx_out, unsigned *seeds_out);
int  dg_oracle_nullspace(double *A, double *ns, int n);
void dg_oracle_slcm(const double *A, double *B, double *p);
int  dg_oracle_rroots3(const double *po, double *r);
int  dg_oracle_eig_sym(double *a, double *w, int n);
int  dg_oracle_svduv(double *d, double *a, double *u, int m, double *v, int n);
int  dg_oracle_minv(double *a, int n);
void dg_oracle_FDs(const double *u, const double *F, double *p, int len);
void dg_oracle_exFDs(const double *u, const double *F, double *p, double *w, int len);
void dg_oracle_FDsSym(const double *u, const double *F, double *p, int len);
void dg_oracle_HDs(const double *u, const double *H, double *p, int len);
void dg_oracle_HDS_full(int kind, const double *u, const double *H, double *p, int len);
void dg_oracle_u2f(const double *u, const int *inl, int len, double *F);
void dg_oracle_u2fw(const double *u, const int *inl, const double *w, int len, double *F);
void dg_oracle_u2h(const double *u, const int *inl, int len, double *H);
dg_score dg_oracle_inlidxs(const double *err, int len, double th, int *inl);
int  dg_oracle_nsamples(int ninl, int ptNum, int samsiz, double conf);
uint32_t dg_oracle_hash(const int *list, int count);
int  dg_oracle_checksample(const double *F, const double *u7, double th, double *H);
int  dg_oracle_all_ori_valid(const double *F, const double *u, const int *idx, int N);

#ifdef __cplusplus
}
#endif
#endif
