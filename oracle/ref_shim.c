/* TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Shim that sits next to the UNMODIFIED reference sources (compiled where they lie under
 * /root/reference by oracle/Makefile) and makes them usable as a deterministic checker:
 *
 *   - oracle_time(): exp_ranF.c / exp_ranH.c are compiled with -Dtime=oracle_time, so the
 *     reference's `srand(time(NULL))` (exp_ranF.c:1277, exp_ranH.c:510) becomes
 *     `srand(oracle_seed)`.  Nothing in the reference is edited.
 *   - ref_find_fundamental / ref_find_homography: the marshalling that
 *     src/pydegensac/bindings.cpp:297-435 and :64-222 do in C++ (build u[N][6] and the two LAF
 *     point sets, pick metric function pointers, threshold conventions), restated in C so that
 *     ctypes can call the reference without pybind11.
 *   - counting thunks for the injected metric pointers (FDS1/EXFDS1/HDS1): number of full
 *     N-point scoring passes ("models scored") and time spent in them; with
 *     count_models they also log (time since the driver started, model) of every pass, so that
 *     ref_time_to_best() can report when the model the driver RETURNS was first scored
 *     (exp_ranF.c:1381-1421 / :1523-1569 commit a model right after that pass) — the
 *     reference-side half of BASELINE.json's "time-to-best-inlier-set".
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "degensac/exp_ranF.h"
#include "degensac/exp_ranH.h"
#include "degensac/Htools.h"
#include "degensac/Ftools.h"

static unsigned g_seed = 1u;
time_t oracle_time(time_t *t) { if (t) *t = (time_t)g_seed; return (time_t)g_seed; }
void oracle_set_seed(unsigned s) { g_seed = s; }

/* ---- counting thunks ---------------------------------------------------------------------- */
static long long g_full_passes = 0;   /* FDS1 / HDS1 calls  (one model scored on all N points) */
static long long g_ex_passes = 0;     /* EXFDS1 calls (LO iterations)                          */
static double    g_pass_seconds = 0;
static int       g_time_passes = 0;

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* pass log: when each scored model was seen (seconds since the driver call started) */
typedef struct { double t; double m[9]; } pass_rec;
static pass_rec *g_log = 0; static size_t g_log_n = 0, g_log_cap = 0; static int g_log_on = 0;
static double g_t_start = 0, g_t_best = -1;
static void log_pass(const double *m) {
    if (!g_log_on) return;
    if (g_log_n == g_log_cap) { g_log_cap = g_log_cap ? 2 * g_log_cap : 4096; g_log = (pass_rec *)realloc(g_log, g_log_cap * sizeof(pass_rec)); }
    g_log[g_log_n].t = now_s() - g_t_start; memcpy(g_log[g_log_n].m, m, 72); g_log_n++;
}
static void log_begin(int on) { g_log_on = on; g_log_n = 0; g_t_best = -1; g_t_start = now_s(); }
/* the first logged pass over exactly the returned model (bitwise); -1 when the driver returns a model it never scored */
static void log_end(const double *model) {
    size_t i;
    g_t_best = -1;
    for (i = 0; i < g_log_n && g_log_on; i++) if (!memcmp(g_log[i].m, model, 72)) { g_t_best = g_log[i].t; break; }
    g_log_on = 0;
}
double ref_time_to_best(void) { return g_t_best; }

static FDsPtr   g_FDS = 0;
static exFDsPtr g_EXFDS = 0;
static HDsPtr   g_HDS = 0;

static void thunk_FDS(const double *u, const double *F, double *p, int len) {
    double t0 = g_time_passes ? now_s() : 0;
    g_FDS(u, F, p, len);
    if (g_time_passes) g_pass_seconds += now_s() - t0;
    g_full_passes++; log_pass(F);
}
static void thunk_EXFDS(const double *u, const double *F, double *p, double *w, int len) {
    double t0 = g_time_passes ? now_s() : 0;
    g_EXFDS(u, F, p, w, len);
    if (g_time_passes) g_pass_seconds += now_s() - t0;
    g_ex_passes++; log_pass(F);
}
static void thunk_HDS(const double *lin, const double *u, const double *H, double *p, int len) {
    double t0 = g_time_passes ? now_s() : 0;
    g_HDS(lin, u, H, p, len);
    if (g_time_passes) g_pass_seconds += now_s() - t0;
    g_full_passes++; log_pass(H);
}

void ref_counters_reset(int time_passes) {
    g_full_passes = 0; g_ex_passes = 0; g_pass_seconds = 0; g_time_passes = time_passes;
}
void ref_counters_get(long long *full, long long *ex, double *seconds) {
    *full = g_full_passes; *ex = g_ex_passes; *seconds = g_pass_seconds;
}

/* ---- marshalling (bindings.cpp:126-198 / :337-409) ----------------------------------------- */
static void build_u(const double *x1, const double *x2, int n, int dim, int laf,
                    double *u, double *ulaf1, double *ulaf2) {
    int i;
    for (i = 0; i < n; i++) {
        const double *a = x1 + (size_t)dim * i, *b = x2 + (size_t)dim * i;
        u[6*i+0] = a[0]; u[6*i+1] = a[1]; u[6*i+2] = 1.;
        u[6*i+3] = b[0]; u[6*i+4] = b[1]; u[6*i+5] = 1.;
        if (laf) {
            /* (x + a12, y + a22) */
            ulaf1[6*i+0] = a[0] + a[3]; ulaf1[6*i+1] = a[1] + a[5]; ulaf1[6*i+2] = 1.;
            ulaf1[6*i+3] = b[0] + b[3]; ulaf1[6*i+4] = b[1] + b[5]; ulaf1[6*i+5] = 1.;
            /* (x + a11, y + a21) */
            ulaf2[6*i+0] = a[0] + a[2]; ulaf2[6*i+1] = a[1] + a[4]; ulaf2[6*i+2] = 1.;
            ulaf2[6*i+3] = b[0] + b[2]; ulaf2[6*i+4] = b[1] + b[4]; ulaf2[6*i+5] = 1.;
        }
    }
}

/* optional capture of the drivers' per-LO residual dump (`resids`, RESIDS_M rows of n per LO run), which the
 * reference's binding frees unseen: set a destination before the call, cleared after it */
static double *g_resids_dst = 0; static int g_resids_cap = 0;
static int *g_dataout_dst = 0; static int g_dataout_len = 0;
void ref_capture_data_out(int *dst, int len) { g_dataout_dst = dst; g_dataout_len = len; }
void ref_capture_resids(double *dst, int cap_runs) { g_resids_dst = dst; g_resids_cap = cap_runs; }
static void capture_resids(const double *resids, int runs, int n)
{
    if (g_resids_dst && resids) {
        int r = runs < g_resids_cap ? runs : g_resids_cap;
        if (r > 0) memcpy(g_resids_dst, resids, sizeof(double) * (size_t)r * RESIDS_M * (size_t)n);
    }
    g_resids_dst = 0; g_resids_cap = 0;
}

/* stats: [0]=samples drawn, [1]=LO runs, [2]=(H) rejected samples, [3]=returned inlier count I */
int ref_find_fundamental(const double *x1, const double *x2, int n, int dim,
                         double px_th, double conf, int max_iters, int error_type,
                         int sym_check, double laf_coef, int degen, unsigned seed,
                         int count_models, double *F, unsigned char *mask, int *stats)
{
    FDsPtr FDS1; exFDsPtr EXFDS1; FDsidxPtr FDSidx1;
    double th = px_th * px_th, sym_th = px_th * px_th * 3.0 * (sym_check ? 1 : 0);
    int laf = laf_coef > 0, ret, I_H = 0, i;
    double *u, *ulaf1, *ulaf2, *resids = 0, HinF[9];
    int *data_out;

    if (error_type == 1) { FDS1 = &FDsSym; EXFDS1 = &exFDsSym; FDSidx1 = &FDsSymidx; }
    else                 { FDS1 = &FDs;    EXFDS1 = &exFDs;    FDSidx1 = &FDsidx;    }
    if (count_models) { g_FDS = FDS1; g_EXFDS = EXFDS1; FDS1 = &thunk_FDS; EXFDS1 = &thunk_EXFDS; }

    u  = (double *)malloc(sizeof(double) * 6 * (size_t)n);
    ulaf1 = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    ulaf2 = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    data_out = (int *)calloc((size_t)n * 18 + 8, sizeof(int));
    build_u(x1, x2, n, dim, laf, u, ulaf1, ulaf2);
    for (i = 0; i < 9; i++) F[i] = 0;

    oracle_set_seed(seed);
    log_begin(count_models);
    ret = exp_ransacFcustomLAF(u, ulaf1, ulaf2, n, th, laf_coef, conf, max_iters, F, mask, data_out,
                               1, 0, &resids, HinF, &I_H, EXFDS1, FDS1, FDSidx1, sym_th, degen);
    log_end(F);
    if (stats) { stats[0] = data_out[0]; stats[1] = data_out[1]; stats[2] = I_H; stats[3] = ret; }
    capture_resids(resids, data_out[1], n);
    if (g_dataout_dst) { memcpy(g_dataout_dst, data_out, sizeof(int) * (size_t)g_dataout_len); g_dataout_dst = 0; }
    free(resids); free(data_out); free(u); free(ulaf1); free(ulaf2);
    return ret;
}

int ref_find_homography(const double *x1, const double *x2, int n, int dim,
                        double px_th, double conf, int max_iters, int error_type,
                        int sym_check, double laf_coef, unsigned seed,
                        int count_models, double *H, unsigned char *mask, int *stats)
{
    HDsPtr HDS1; HDsiPtr HDSi1; HDsidxPtr HDSidx1;
    double th, sym_th, coef = 3.0 * (sym_check ? 1 : 0);
    int laf = laf_coef > 0, i;
    double *u, *ulaf1, *ulaf2, *resids = 0;
    int *data_out;
    Score S;

    switch (error_type) {
    case 1:  HDS1 = &HDsSymMaxSq; HDSi1 = &HDsiSymMaxSq; HDSidx1 = &HDsSymMaxSqidx; th = px_th*px_th; sym_th = 0; break;
    case 2:  HDS1 = &HDsSymMax;   HDSi1 = &HDsiSymMax;   HDSidx1 = &HDsSymMaxidx;   th = px_th;       sym_th = 0; break;
    case 3:  HDS1 = &HDsSymSumSq; HDSi1 = &HDsiSymSumSq; HDSidx1 = &HDsSymSumSqidx; th = px_th*px_th; sym_th = px_th*coef; break;
    case 4:  HDS1 = &HDsSymSum;   HDSi1 = &HDsiSymSum;   HDSidx1 = &HDsSymSumidx;   th = px_th;       sym_th = px_th*coef; break;
    default: HDS1 = &HDs;         HDSi1 = &HDsi;         HDSidx1 = &HDsidx;         th = px_th*px_th; sym_th = px_th*coef; break;
    }
    if (count_models) { g_HDS = HDS1; HDS1 = &thunk_HDS; }

    u  = (double *)malloc(sizeof(double) * 6 * (size_t)n);
    ulaf1 = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    ulaf2 = (double *)malloc(sizeof(double) * 6 * (size_t)(laf ? n : 1));
    data_out = (int *)calloc((size_t)n * 18 + 8, sizeof(int));
    build_u(x1, x2, n, dim, laf, u, ulaf1, ulaf2);
    for (i = 0; i < 9; i++) H[i] = 0;

    oracle_set_seed(seed);
    log_begin(count_models);
    S = exp_ransacHcustomLAF(u, ulaf1, ulaf2, n, th, laf_coef, conf, max_iters, H, mask, 4, data_out,
                             1, 0, &resids, HDS1, HDSi1, HDSidx1, sym_th);
    log_end(H);
    if (stats) { stats[0] = data_out[0]; stats[1] = data_out[1]; stats[2] = data_out[2]; stats[3] = (int)S.I; }
    capture_resids(resids, data_out[1], n);
    free(resids); free(data_out); free(u); free(ulaf1); free(ulaf2);
    return (int)S.I;
}

/* ---- the legacy fundamental-matrix drivers of exp_ranF.c: exp_ransacFcustom (:811, the custom-metric
 * driver without the LAF arguments) and exp_ransacF (:242, fixed Sampson metric, never seeds the generator itself).
 * Same marshalling as ref_find_fundamental; variant 0 = exp_ransacFcustom, 1 = exp_ransacF. */
/* defined in exp_ranF.c:811 but not declared in exp_ranF.h */
int exp_ransacFcustom(double *u, int len, double th, double conf, int max_sam, double *F, unsigned char *inl, int *data_out,
                      int do_lo, unsigned inlLimit, double **resids, double *H_best, int *Ih, exFDsPtr EXFDS1, FDsPtr FDS1, int doSymCheck);
int ref_find_fundamental_legacy(int variant, const double *x1, const double *x2, int n, int dim,
                                double px_th, double conf, int max_iters, int error_type, int sym_check, unsigned seed,
                                double *F, unsigned char *mask, int *stats)
{
    FDsPtr FDS1; exFDsPtr EXFDS1;
    double th = px_th * px_th;
    int ret, I_H = 0, i;
    double *u, *ua, *ub, *resids = 0, HinF[9];
    int *data_out;
    if (error_type == 1) { FDS1 = &FDsSym; EXFDS1 = &exFDsSym; } else { FDS1 = &FDs; EXFDS1 = &exFDs; }
    u = (double *)malloc(sizeof(double) * 6 * (size_t)n); ua = (double *)malloc(48); ub = (double *)malloc(48);
    data_out = (int *)calloc((size_t)n * 18 + 8, sizeof(int));
    build_u(x1, x2, n, dim, 0, u, ua, ub);
    for (i = 0; i < 9; i++) F[i] = 0;
    oracle_set_seed(seed);
    if (variant == 0)
        ret = exp_ransacFcustom(u, n, th, conf, max_iters, F, mask, data_out, 1, 0, &resids, HinF, &I_H, EXFDS1, FDS1, sym_check);
    else {
        srand(seed);                       /* exp_ransacF draws `seed = rand()` from whatever state the process is in */
        ret = exp_ransacF(u, n, th, conf, max_iters, F, mask, data_out, 1, 0, &resids, HinF, &I_H);
    }
    if (stats) { stats[0] = data_out[0]; stats[1] = data_out[1]; stats[2] = I_H; stats[3] = ret; }
    free(resids); free(data_out); free(u); free(ua); free(ub);
    return ret;
}

/* ---- ransacH2el (ranH2el.c:19): no binding in the reference's Python layer, called as its header declares
 * it.  The driver draws `seed = rand()` from the process-wide generator without seeding it: srand(seed) here. */
/* ranH2el.h:35 (its header does not compile on its own after Ftools.h: macro clashes) */
Score ransacH2el(double *u10, int len, double th, double conf, int max_sam, double *H, unsigned char *inl, int *data_out, int do_lo, int inlLimit);
int ref_ransacH2el(const double *u10, int n, double th, double conf, int max_iters, int do_lo, int inl_limit, unsigned seed,
                   double *H, unsigned char *mask, int *stats)
{
    int data_out[3] = {0, 0, 0}, i; Score S;
    double *u = (double *)malloc(sizeof(double) * 10 * (size_t)n);
    memcpy(u, u10, sizeof(double) * 10 * (size_t)n);
    for (i = 0; i < 9; i++) H[i] = 0;
    srand(seed);
    S = ransacH2el(u, n, th, conf, max_iters, H, mask, data_out, do_lo, inl_limit);
    if (stats) { stats[0] = data_out[0]; stats[1] = data_out[1]; stats[2] = data_out[2]; stats[3] = (int)S.I; }
    free(u);
    return (int)S.I;
}
