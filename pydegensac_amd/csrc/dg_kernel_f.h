/* Persistent fundamental-matrix kernel: one workgroup runs the whole reference driver
 * exp_ransacFcustomLAF (degensac/exp_ranF.c:1244-1767) for one image pair.
 *
 * Main loop = speculate-then-commit: the sample stream does not depend on scoring
 * outcomes, so DG_CHUNK minimal samples are drawn (lane-parallel glibc replay + sequential pool
 * swaps), solved (one 7-point problem per lane, registers only) and scored (one wave per model, the
 * point set resident in LDS), then a workgroup-uniform commit scan replays exp_ranF.c:1365-1577 in
 * order and fires the rare heavy branches (symmetric/LAF checks, DEGENSAC, local optimisation) as
 * cooperative passes.  Residual vectors are never materialised: every `errs[k]` buffer of the
 * reference is tracked as "the model whose residuals it would hold" and recomputed on demand.
 */
#ifndef DG_KERNEL_F_H
#define DG_KERNEL_F_H
#include <stddef.h>
#include "dg_lsq.h"

struct dg_score { unsigned I; double J; unsigned Is; unsigned Ilafs; };

/* doubles that ww[] lacks for the parallel pool stage's touch table */
#define DG_WPAD ((2 * DG_CHUNK * 7 * 4 > DG_NW * (int)sizeof(dg_wave_ws)) ? (2 * DG_CHUNK * 7 * 4 - DG_NW * (int)sizeof(dg_wave_ws) + 7) / 8 : 1)
/* the F driver's workgroup-uniform state between two chunks (replicated in every lane while a pair runs; one copy in
 * LDS travels with the image of a pair that is set aside, dg_args::park_sam) */
struct dg_f_drv {
    dg_score maxS, maxSs;
    int no_sam, max_sam, iter_cnt, degen_cnt, iterID, Ihmax; unsigned non_degen;
    int best_sample; long long t_best, t_start;
    long long t_parked;              /* device clock when the pair was set aside (its waiting time is taken out of the reported times) */
    int flast_k, has_last;           /* legacy drivers' symmetric check: sample whose last model is in flast (0 = none); lastIds valid */
    int finKind, accepted, perm[4], p4, e4kind, track, done;
    unsigned seed; int cur, chunk_s[3], chunk_base;
    int n_fds, n_exfds, n_hds, n_aux;
};

/* A 14-point sample of the local optimisation prepared ahead of time by an idle wave (dg_inFrani): the generator state
 * it assumes at the start of the repetition, the state after its draws, the list slots the draws would store, the model. */
#ifndef DG_LO_AHEAD_MAX
#define DG_LO_AHEAD_MAX 5
#endif
#define DG_LO_AHEAD (DG_NW - 1 < DG_LO_AHEAD_MAX ? DG_NW - 1 : DG_LO_AHEAD_MAX)
struct dg_lo_ahead { dg_rng before, after; int pos[32], val[32]; double F[9]; };

/* LDS scratch of the long-list least squares (640 doubles): the per-wave solver scratch, idle while a workgroup fit runs */
#define DG_LSQ_LTAB(S) (sizeof((S)->ww) + sizeof((S)->wpad) >= 640 * sizeof(double) ? (double *)(S)->ww : (double *)0)

/* The pair's workspace views (workgroup-uniform pointers).  They live in LDS (dg_f_shared::K) and are read where they
 * are used, instead of travelling through the whole driver in every lane's vector registers. */
struct dg_f_cshared {
    int *L[10];              /* global int lists: 0 inliers, 1 intbuff, 2 intbuff_best (LO); 3 inliersH, 4 intbuffH (innerH);
                                5 idxN, 6 idxH, 7 idxV, 8 ptr (rFtH); 9 inlI (u2Fit) */
    unsigned char *Fl[5];    /* global flag vectors: 0 hinl, 1 nhinl, 2 vN, 3 v (innerFH), 4 inl (innerFH result) */
    double *gmodels;         /* [3*DG_CHUNK][9] chunk models */
    dg_pt *stage;            /* [2 * n_max] gathered correspondences of a long least-squares list + its per-coordinate arrays */
    unsigned *res_I; double *res_J;   /* [3*DG_CHUNK] per-model (I, J) of the current chunk */
    int (*rf)[5];            /* [DG_CHUNK] rFtH batch: candidate point ids (2), swap log (2), count */
    int *wlist; dg_pt *wstage;        /* per-wave buffers [DG_NW][n_max] of the wave-parallel innerFH / u2Fit */
    char *hrep;              /* homography LO with one repetition per wave (dg_ws_layout::off_hrep), or null */
    int n_max;
};

/* innerH with one repetition per wave (dg_innerH_waves): what a repetition starts from (its sample, the generator right after the
 * sample's draws) and what it leaves (the generator after its own draws, their number, the best model of the repetition and
 * its MSAC gain, the passes it made) */
struct dg_ih_log { dg_rng g; double h[9]; double itJ; int ids[12]; int passes, draws;
                   int pub, aborted; };    /* pub: -1 while it runs, then its draws (the later repetitions of the round watch it); aborted: stopped as stale */

/* The fundamental-matrix local optimisation with one repetition per wave (dg_inFrani_waves): what a repetition starts from
 * (sample ids, the generator right after the sample's 14 draws) and what it leaves for the in-order replay: per iteration of
 * exp_iterFcustom the (hash, I) of its inlier set and whether the re-fit behind it drew an 8-subset; its own result. */
struct dg_lo_it { unsigned hash; int I; int drew; };
struct dg_lo_log {
    dg_rng g, g0;                         /* g0: the generator right behind the sample's draws (g moves on with the repetition's own draws) */
    /* the list slots the sample's draws stored, with the values they replaced (-1: none): undone when the repetition is not committed */
    int upos[28], uval[28];
    int ids[14];
    /* r0.I (< 8: the repetition ends at once), draws of the first 8-subset, iterations that hashed their set, final pass made */
    int I0, drew0, nit, has_fin;
    dg_lo_it it[DG_ILSQ_ITERS];
    double f[9], J; int I, kind0;         /* result when no iteration is cut short by the replay: model, score, metric variant of errs[0] */
    int cut, draws, n_ex, n_fd;           /* filled by the replay: cut short by an earlier repetition's set, 8-subset draws really consumed, passes to count */
    /* pub: -1 while the repetition runs, then the draws it made (the later repetitions of the round watch it); aborted: stopped as stale */
    int pub, aborted;
};
/* cooperative large-n mode: the repetitions of a round are units of stage 4, one claiming workgroup each; their records live in
 * the owner's workspace (DG_LOJOB_BYTES at the end of the stage-3 staging): a header, then one record per repetition on its own
 * 128-byte lines (records are written by workgroups on different XCDs) */
#define DG_LOJOB_STRIDE 768
#define DG_LOJOB_BYTES (128 + DG_RAN_REP * DG_LOJOB_STRIDE)
static_assert(sizeof(dg_lo_log) <= DG_LOJOB_STRIDE, "dg_lo_log does not fit its slot");
struct dg_lo_job { int n, ssiz, mk_full, mk_ex; double th; };

struct dg_f_shared {
    dg_red red;
    dg_lsq_scratch lsq;
    dg_rng rng, rng_save;
    /* triple-buffered chunk state: while chunk c is scored, wave 0 runs the pool swaps of chunk c+1 and
     * wave 1 the seed chain + raw draws of chunk c+2 (the sample stream does not depend on outcomes) */
    unsigned seeds3[3][DG_CHUNK];
    int      draws3[3][DG_CHUNK][8];    /* raw draws, then drawn ids (draw order) */
    unsigned long long alm3[3][DG_CHUNK / 64];   /* per-sample alias flags (order-dependent swaps) */
    dg_wave_ws ww[DG_NW];                /* per-wave scratch of the wave-parallel sections */
    double   wpad[DG_WPAD];              /* extends ww[] to the size the parallel pool stage needs (2 * DG_CHUNK * 7 ints) */
    double   csH[5][9]; int csRes[5];    /* checksample: per-triplet homography and verdict */
    union {      /* never live at the same time: innerFH and innerH are separate steps of the DEGENSAC branch, `ahead` belongs to one local optimisation */
        struct {
            double   fhF[16][9], fhF2[16][9], fhTh[16];   /* innerFH: per-repetition model, its u2Fit refinement, flag threshold */
            int      fhIds[16][10], fhCnt[16], fhCnt2[16], fhRaw[160];
        };
        dg_lo_ahead ahead[DG_LO_AHEAD];
        struct { dg_ih_log ih[DG_NW]; dg_rng ih_start, ih_work; };    /* innerH, one repetition per wave */
        struct { dg_lo_log lo[DG_NW]; dg_rng lo_start, lo_work; };    /* local optimisation, one repetition per wave */
    };
    int n_ahead;
    int n_lafrej;                        /* candidates the LAF check turned down (thread 0 counts; MI_ST_REJECTED of the F driver) */
    unsigned scnt[4];                    /* main-loop models by the arithmetic they got (dg_score_chunk_F): [0] entered the level-1 screen (fp32, packed),
                                            [1] entered level 2 (fp64, the point's own denominator), [2] scored with the exact metric, [3] all models scored */
    long long ph[8], dbg[8], tq;
#ifdef DG_LO_PROF
    long long lt[16], ltq;
#endif          /* phase timers (lane 0), 100 MHz ticks */
    unsigned short moff[DG_T + 1];      /* first model slot of each sample */
    unsigned char  nv[DG_CHUNK];        /* valid models per sample; 255 = nullspace dimension != 2 */
    unsigned char  ridx[DG_CHUNK][4];   /* root index i (= errs[] slot) of each valid model */
    unsigned short mslot[3 * DG_CHUNK]; /* compact (ordered) model index -> slot in the chunk's model table */
    unsigned wave_cnt[DG_NW];
    double   f[9], F[9], FBest[9], H[9], Hx[9], fLO[9], ftmp[9];
    double   bufF[4][9]; int bufKind[4]; /* model whose residuals each physical errs[] buffer holds */
    double   u7[7][4];
    double   ext[4], extw[DG_NW][4];     /* max |x1|, |y1|, |x2|, |y2| over the pair (screening bound), per-wave partials */
    int      samidxBest[7];
    int      itmp[32];
    unsigned char nsolv[DG_CHUNK];       /* real roots of each sample's cubic (legacy drivers' symmetric check) */
    double   flast[9]; int lastIds[8];    /* ... the model the reference's local `f` holds, the ids of the last sample that reached the cubic */
    double   dtmp[32];
    dg_f_drv park;
    dg_f_cshared K;                      /* the pair's workspace views (dg_f_ctx::K) */
};

/* stream mode: one chunk as the producer workgroup leaves it in the owner's ring.  Past the first samples nearly every model
 * of a chunk fails the screens, so the entry is compact: the sample stream itself (seeds, drawn ids: what an event of the commit
 * re-seeds from and gathers, and what the owner needs to solve and score the chunk itself should the bound have fallen), the
 * number of models of every sample, and the few models whose MSAC gain exceeds the bound the producer used (with more than
 * DG_STREAM_EV_MAX of them — the first chunks of a pair — `overflow` tells the owner to solve and score the chunk itself). */
struct dg_stream_ev { double model[9]; double J; unsigned I; short k; unsigned char r, pad; unsigned char ridx[4]; };
#define DG_STREAM_EV_MAX 40
struct dg_stream_ent {
    int cn, Mtot, n_ev, overflow; double tau_used, pad1;
    unsigned seeds[DG_CHUNK]; int draws[DG_CHUNK][8];
    unsigned char nv[DG_CHUNK];
    dg_stream_ev ev[DG_STREAM_EV_MAX];
};

/* ------------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------------ */
template <int LDSPTS>
struct dg_f_ctx {
    dg_f_shared *S;
    const __attribute__((address_space(3))) dg_f_cshared *K;    /* = &S->K */
    const dg_pt *P;          /* correspondences (LDS or global) */
    int *pool;               /* sampler permutation pool (LDS or global) */
    int n, tid;
    const dg_args *A;
    long long off;           /* first row of this pair in the input arrays */
    dg_ht ht;
    unsigned *seeds; int (*draws)[8]; /* the chunk buffers of the chunk being committed */
    /* counters */
    int n_fds, n_exfds, n_hds, n_aux;
    double *rrun;            /* diagnostics: the 62 x n residual rows of the current LO run, or null */
    dg_coop_cb *cb; int *coop_gen; int coop_slot;   /* cooperative large-n mode: this owner's control block (null = off) */
    double *hlt;             /* homography kernel: [DG_NW][DG_HLT] doubles of LDS, one block per wave (one-repetition-per-wave LO) */
    int hjob_gen;            /* homography kernel: generation of this slot's last local-optimisation job (dg_hjob_cb) */
    int lo_assumed, lo_prev;
        /* cooperative large-n mode: 8-subset draws the last committed repetition of a local optimisation consumed (the start states of the next ones assume it)
         */

    __device__ __forceinline__ dg_pt pt(int i) const { return P[i]; }
    /* LAF point sets u_1 (which=1: +a12,+a22) and u_2 (which=2: +a11,+a21): bindings.cpp:337-409 */
    __device__ __forceinline__ dg_pt laf_pt(int i, int which) const {
        const double *a = A->pts1 + (size_t)(off + i) * 6, *b = A->pts2 + (size_t)(off + i) * 6;
        dg_pt p;
        if (which == 1) { p.x1 = a[0] + a[3]; p.y1 = a[1] + a[5]; p.x2 = b[0] + b[3]; p.y2 = b[1] + b[5]; }
        else            { p.x1 = a[0] + a[2]; p.y1 = a[1] + a[4]; p.x2 = b[0] + b[2]; p.y2 = b[1] + b[4]; }
        return p;
    }
};

/* ww[] and wpad[] are adjacent double arrays that only the single-wave solver sections use: during a workgroup pass
 * they hold the ordered MSAC terms, during the main loop's scoring phase the parallel pool stage's touch table */
#define DG_JBUF_LDS_BYTES (offsetof(dg_f_shared, wpad) + sizeof(((dg_f_shared *)0)->wpad) - offsetof(dg_f_shared, ww))
static_assert(offsetof(dg_f_shared, wpad) == offsetof(dg_f_shared, ww) + sizeof(((dg_f_shared *)0)->ww), "ww and wpad must be contiguous");
static_assert(DG_JBUF_LDS_BYTES >= 2 * DG_CHUNK * 7 * sizeof(int), "pool-stage scratch does not fit");

/* thread 0 fills the pair's workspace views (dg_f_shared::K); the caller's next workgroup barrier publishes them */
__device__ __forceinline__ void dg_fill_views(dg_f_cshared *K, char *ws, const dg_ws_layout &wl)
{
    for (int i = 0; i < 10; i++) K->L[i] = (int *)(ws + wl.off_lists) + (size_t)i * wl.n_max;
    for (int i = 0; i < 5; i++) K->Fl[i] = (unsigned char *)(ws + wl.off_flags) + (size_t)i * wl.n_max;
    K->gmodels = (double *)(ws + wl.off_models);
    K->stage = (dg_pt *)(ws + wl.off_stage);
    K->res_J = (double *)(ws + wl.off_res); K->res_I = (unsigned *)(K->res_J + 3 * DG_CHUNK); K->rf = (int (*)[5])(K->res_I + 3 * DG_CHUNK);
    K->hrep = wl.hrep ? ws + wl.off_hrep : (char *)0;
    K->n_max = wl.n_max; K->wlist = (int *)(ws + wl.off_wave); K->wstage = (dg_pt *)(ws + wl.off_wave + (size_t)DG_NW * wl.n_max * sizeof(int));
}

#define CTX dg_f_ctx<LDSPTS>
#define DG_RESIDS_M 62           /* rtools.h:15: 2 + RAN_REP * (1 + ILSQ_ITERS + 1) residual vectors per LO run */

/* diagnostics: row `row` of the current LO run's dump := residuals of model M (LDS) under metric `kind`
 * (kind < 10: fundamental-matrix metrics of dg_Ferr; >= 10: homography metric kind - 10) */
template <int LDSPTS>
__device__ __forceinline__ void dg_dump_resid(CTX &c, int row, const double *M, int kind)
{
    if (!c.rrun) return;
    double m[9], Hinv[9], H1[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { m[i] = M[i]; Hinv[i] = 0; H1[i] = 0; }
    if (kind > 10) dg_hsym_prepare(m, Hinv, H1);
    double *out = c.rrun + (size_t)row * c.n;
    for (int j = c.tid; j < c.n; j += DG_T) {
        const dg_pt q = dg_ldpt<LDSPTS>(c.P, j);
        out[j] = kind < 10 ? dg_Ferr(kind, m, q) : dg_Herr(kind - 10, m, Hinv, H1, q);
    }
}
/* a new LO run: point c.rrun at its rows (null when the dump is off or full) and mark every row "never written" */
template <int LDSPTS>
__device__ __forceinline__ void dg_resid_begin(CTX &c, int run /* 0-based */)
{
    c.rrun = 0;
    if (!c.A->resids_out || run >= c.A->resid_runs) return;
    c.rrun = c.A->resids_out + ((size_t)c.off * c.A->resid_runs + (size_t)run * c.n) * DG_RESIDS_M;
    const double nan_ = __longlong_as_double(0x7ff8000000000000ll);
    for (size_t j = c.tid; j < (size_t)DG_RESIDS_M * c.n; j += DG_T) c.rrun[j] = nan_;
}

/* debug checkpoints (tag, I, J), mirrored by the oracle's TRACE2 hook; active only when A.trace != 0 */
#define DG_TRACE(c, tag, I, J) do { if ((c).A->trace && (c).tid == 0) { int *t_ = (c).A->trace; int k_ = t_[0]; \
    if (k_ < (c).A->trace_cap) { long long jb_ = __double_as_longlong((double)(J)); t_[1+4*k_] = (tag); t_[2+4*k_] = (int)(I); \
    t_[3+4*k_] = (int)(jb_ & 0xffffffffll); t_[4+4*k_] = (int)(jb_ >> 32); t_[0] = k_ + 1; } } } while (0)

/* a full scoring pass of model F (kind) with optional list/flags; counts as FDS1/EXFDS1/aux */
template <int LDSPTS>
__device__ __noinline__ dg_pass_res dg_coop_pass(CTX &c, const double *Fm /* LDS */, int kind, const dg_pass_cfg &cfg);

template <int LDSPTS>
__device__ __forceinline__ dg_pass_res dg_f_pass(CTX &c, const double *Fm /* LDS */, int kind, dg_pass_cfg cfg)
{
    /* cooperative large-n mode: passes over the whole point set are distributed over the claiming workgroups */
    if (LDSPTS == 0 && c.cb && !cfg.src && !cfg.flags && !cfg.wantC && cfg.n >= c.A->coop_pass_min) return dg_coop_pass<LDSPTS>(c, Fm, kind, cfg);
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; i++) F[i] = Fm[i];
    const dg_pt *P = c.P;
    /* ordered MSAC terms: the per-wave solver scratch is idle during a workgroup pass (LDS); what does not fit goes to the HBM staging area */
    cfg.jbuf = (double *)c.K->stage; cfg.jl = (double *)c.S->ww; cfg.jl_cap = (int)(DG_JBUF_LDS_BYTES / sizeof(double));
    return dg_pass(&c.S->red, cfg, [&](int pid, int) { return dg_Ferr(kind, F, dg_ldpt<LDSPTS>(P, pid)); }, c.tid);
}
template <int LDSPTS>
__device__ __forceinline__ dg_pass_res dg_h_pass(CTX &c, const double *Hm /* LDS */, dg_pass_cfg cfg)
{
    double H[9];
#pragma unroll
    for (int i = 0; i < 9; i++) H[i] = Hm[i];
    const dg_pt *P = c.P;
    /* ordered MSAC terms: the per-wave solver scratch is idle during a workgroup pass (LDS); what does not fit goes to the HBM staging area */
    cfg.jbuf = (double *)c.K->stage; cfg.jl = (double *)c.S->ww; cfg.jl_cap = (int)(DG_JBUF_LDS_BYTES / sizeof(double));
    return dg_pass(&c.S->red, cfg, [&](int pid, int) { dg_pt p = dg_ldpt<LDSPTS>(P, pid); return dg_HDs(H, p.x1, p.y1, p.x2, p.y2); }, c.tid);
}
__device__ __forceinline__ dg_pass_cfg dg_cfg0(int n)
{
    dg_pass_cfg c; c.n = n; c.src = 0; c.p0 = 0; c.wantJ = 0; c.thJ = 0; c.jbuf = 0; c.jl = 0; c.jl_cap = 0; c.wantC = 0; c.thC = 0; c.list = 0; c.thL = 0;
        c.listStrict = 0; c.list2 = 0; c.thL2 = 0; c.flags = 0; c.thF = 0;
    return c;
}

/* gather `len` points of a global id list into the lane-0 scratch (coordinates x1,y1,x2,y2) */
template <int LDSPTS>
__device__ __forceinline__ void dg_gather(CTX &c, const int *ids, int len, double *px)
{
    for (int i = 0; i < len; i++) { dg_pt p = dg_ldpt<LDSPTS>(c.P, ids[i]); px[4*i] = p.x1; px[4*i+1] = p.y1; px[4*i+2] = p.x2; px[4*i+3] = p.y2; }
}

/* lane j < len writes the coordinates of point `id` (its own) to row j of px */
template <int LDSPTS>
__device__ __forceinline__ void dg_gather_wave(CTX &c, int id, int len, double *px, int lane)
{
    if (lane < len) { dg_pt p = dg_ldpt<LDSPTS>(c.P, id); px[4*lane] = p.x1; px[4*lane+1] = p.y1; px[4*lane+2] = p.x2; px[4*lane+3] = p.y2; }
}

/* u2f on a global id list of any length -> S->f  (exp_ranF.c's u2f(u, inliers, n, f, buffer) calls) */
template <int LDSPTS>
__device__ __forceinline__ void dg_u2f_list(CTX &c, const int *list, int len, const double *wmodel, int wkind, double *Fout)
{
    dg_f_shared *S = c.S;
    if (len <= 16) {
        __syncthreads();
        if (c.tid < 64) {
            const int lane = c.tid;
            dg_gather_wave(c, lane < len ? list[lane] : 0, len, S->lsq.px, lane);
            if (wmodel && lane < len) {
                dg_pt q = dg_ldpt<LDSPTS>(c.P, list[lane]);
                if (wkind == DG_K_FDS) S->lsq.part[0][lane] = dg_exFDs_w(wmodel, q.x1, q.y1, q.x2, q.y2);
                else { double w; dg_exFDsSym(wmodel, q.x1, q.y1, q.x2, q.y2, &w); S->lsq.part[0][lane] = w; }
            }
            DG_WSYNC();
            dg_u2f_small_w(&S->lsq, S->lsq.px, wmodel ? S->lsq.part[0] : 0, len, Fout, c.tid);
        }
        __syncthreads();
    } else {
        const dg_pt *P = c.P;
        dg_u2f_big(&S->red, &S->lsq, [&](int i) { return dg_ldpt<LDSPTS>(P, i); }, list, len, c.tid, Fout, c.K->stage, 2 * c.K->n_max,
                   DG_LSQ_LTAB(S));
    }
}

/* symmetric + LAF consistency of candidate f over the ids list[0..cnt) (exp_ranF.c:1383-1411,
 * :1526-1556, :1654-1682).  Returns 0 when the candidate must be rejected. */
template <int LDSPTS>
__device__ __forceinline__ int dg_f_checks(CTX &c, const double *f, const int *list, int cnt, dg_score &S, const dg_score &maxS, int mkind)
{
    const dg_params &pr = c.A->prm;
    if (pr.sym_th > 0) {
        /* exp_ransacFcustom counts the symmetric-consistent points over ALL points (exp_ranF.c:943-953), the LAF driver over
         * the candidate's inliers */
        dg_pass_cfg cfg = pr.legacy ? dg_cfg0(c.n) : dg_cfg0(cnt); if (!pr.legacy) cfg.src = list; cfg.wantC = 1; cfg.thC = pr.sym_th;
        dg_pass_res r = dg_f_pass(c, f, DG_K_FSYM, cfg);
        S.Is = r.C;
        if (S.Is < maxS.Is) return 0;
    }
    if (pr.laf_coef > 0) {
        double thl = pr.laf_coef * pr.th;
        double F[9];
        for (int i = 0; i < 9; i++) F[i] = f[i];
        dg_pass_cfg cfg = dg_cfg0(cnt); cfg.src = list; cfg.wantC = 1; cfg.thC = thl;
        dg_pass_res r1 = dg_pass(&c.S->red, cfg, [&](int pid, int) { return dg_Ferr(mkind, F, c.laf_pt(pid, 1)); }, c.tid);
        dg_pass_res r2 = dg_pass(&c.S->red, cfg, [&](int pid, int) { return dg_Ferr(mkind, F, c.laf_pt(pid, 2)); }, c.tid);
        S.Ilafs = r2.C < r1.C ? r2.C : r1.C;
        if (S.Ilafs < maxS.Ilafs) { if (c.tid == 0) c.S->n_lafrej++; return 0; }
    }
    return 1;
}

/* DegUtils.c:42-82 checksample.  The five triplets are independent until the "first success wins" rule:
 * waves 0..4 each evaluate one (Hdetect + sort on lane 0, the 5-point re-fit wave-cooperatively), then the
 * lowest successful index is taken — the same H the sequential loop returns.  Called by the whole workgroup. */
template <int LDSPTS>
__device__ __noinline__ int dg_checksample(CTX &c, const double *F /* LDS */, const double (*u7)[4] /* LDS */, double th, double *H /* LDS out */)
{
    dg_f_shared *S = c.S; const int tid = c.tid, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
#if DG_NW >= 5
    for (int tr = wave; tr < 5; tr += DG_NW) {
        dg_wave_ws *w = &S->ww[wave];
        const unsigned char IDXS[5][3] = {{0,1,2}, {3,4,5}, {0,1,6}, {3,4,6}, {2,5,6}};
        if (lane == 0) {
            dg_Hdetect(F, u7, IDXS[tr], w->H);
            for (int j = 0; j < 7; j++) { w->Ds[j] = dg_HDs(w->H, u7[j][0], u7[j][1], u7[j][2], u7[j][3]); w->sDs[j] = w->Ds[j]; w->idx[j] = j; }
            for (int a = 0; a < 7; ++a)                                  /* sortDs, DegUtils.c:164-183 */
                for (int b = a + 1; b < 7; ++b)
                    if (w->sDs[b] < w->sDs[a]) { double t = w->sDs[b]; w->sDs[b] = w->sDs[a]; w->sDs[a] = t; int ti = w->idx[b]; w->idx[b] = w->idx[a];
                        w->idx[a] = ti; }
            for (int j = 0; j < 5; ++j) { const double *q = u7[w->idx[j]]; w->cpx[4*j] = q[0]; w->cpx[4*j+1] = q[1]; w->cpx[4*j+2] = q[2]; w->cpx[4*j+3] = q[3];
                }
        }
        DG_WSYNC();
        dg_u2h_norm_w(w, w->cpx, 5, w->H, lane);
        if (lane == 0) {
            int inlCount = 0;
            for (int j = 0; j < 7; ++j) if (dg_HDs(w->H, u7[j][0], u7[j][1], u7[j][2], u7[j][3]) < th) ++inlCount;
            S->csRes[tr] = inlCount > 4;
            for (int j = 0; j < 9; j++) S->csH[tr][j] = w->H[j];
        }
        DG_WSYNC();
    }
#else
    /* fewer waves than triplets: a wave takes two triplets at a time, one per half-wave (dg_fit_norm_w2: both eigen-solves in one pass), the
     * second one's scratch in the workgroup's least-squares block, which is idle here.  Four waves: one round instead of two. */
    static_assert(sizeof(dg_lsq_scratch) >= (size_t)DG_NW * DG_X2_DOUBLES * sizeof(double), "second-problem scratch");
    for (int base = wave; base < 5; base += 2 * DG_NW) {
        dg_wave_ws *w = &S->ww[wave];
        double *x2 = (double *)&S->lsq + (size_t)wave * DG_X2_DOUBLES;
        const bool uph = lane >= 32, two = base + DG_NW < 5, mine = (lane & 31) == 0 && (!uph || two);
        const int tr = uph ? base + DG_NW : base;
        double *Hh = uph ? x2 + DG_X2_USER : w->H, *Ds = uph ? x2 + DG_X2_USER + 9 : w->Ds, *sDs = uph ? x2 + DG_X2_USER + 16 : w->sDs;
        int *idx = uph ? (int *)(x2 + DG_X2_USER + 23) : w->idx;
        double *cpx = uph ? x2 + DG_X2_USER + 27 : w->cpx;
        const unsigned char IDXS[5][3] = {{0,1,2}, {3,4,5}, {0,1,6}, {3,4,6}, {2,5,6}};
        if (mine) {
            dg_Hdetect(F, u7, IDXS[tr], Hh);
            for (int j = 0; j < 7; j++) { Ds[j] = dg_HDs(Hh, u7[j][0], u7[j][1], u7[j][2], u7[j][3]); sDs[j] = Ds[j]; idx[j] = j; }
            for (int a = 0; a < 7; ++a)                                  /* sortDs, DegUtils.c:164-183 */
                for (int b = a + 1; b < 7; ++b)
                    if (sDs[b] < sDs[a]) { double t = sDs[b]; sDs[b] = sDs[a]; sDs[a] = t; int ti = idx[b]; idx[b] = idx[a]; idx[a] = ti; }
            for (int j = 0; j < 5; ++j) { const double *q = u7[idx[j]]; cpx[4*j] = q[0]; cpx[4*j+1] = q[1]; cpx[4*j+2] = q[2]; cpx[4*j+3] = q[3]; }
        }
        DG_WSYNC();
        dg_fit_norm_w2<true>(w, x2, w->cpx, x2 + DG_X2_USER + 27, 5, w->H, two ? x2 + DG_X2_USER : (double *)0, lane);
        if (mine) {
            int inlCount = 0;
            for (int j = 0; j < 7; ++j) if (dg_HDs(Hh, u7[j][0], u7[j][1], u7[j][2], u7[j][3]) < th) ++inlCount;
            S->csRes[tr] = inlCount > 4;
            for (int j = 0; j < 9; j++) S->csH[tr][j] = Hh[j];
        }
        DG_WSYNC();
    }
#endif
    __syncthreads();
    int win = -1;
    for (int i = 4; i >= 0; i--) if (S->csRes[i]) win = i;
    if (win >= 0 && tid < 9) H[tid] = S->csH[win][tid];
    __syncthreads();
    return win >= 0;
}

/* ---- ranH.c:18-135 + DegUtils.c:693-731: LO of the plane homography (innerH) -------------------- */
template <int LDSPTS>
__device__ __noinline__ unsigned dg_innerH_serial(CTX &c, double *H /* LDS, in/out */, double th, unsigned inlLimit, unsigned char *inl_flags)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid;
    int *inliers = c.K->L[3], *intbuff = c.K->L[4];
    double *h = S->Hx, *hbest = S->ftmp;            /* hbest = model in errs[0]-chain; H itself is the running best */
    /* d = HDs(H); S = inlidxs(d, th, inliers) */
    dg_pass_cfg cfg = dg_cfg0(n); cfg.list = inliers; cfg.thL = th;
    dg_pass_res r0 = dg_h_pass(c, H, cfg); c.n_hds++;
    int ninl = (int)r0.nL;
    if (ninl >= 8) {                                  /* inHrani, ranH.c:88-135 */
        int ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
        double maxJ = 0;                               /* maxS = {0,0} */
        for (int rep = 0; rep < DG_RAN_REP; ++rep) {
            __syncthreads();
            if (tid < 64) {
                { int id; dg_randsubset_wave(&S->rng, inliers, ninl, ssiz, tid, &id); dg_gather_wave(c, id, ssiz, S->lsq.px, tid); }
                DG_WSYNC();
                dg_u2h_small_w(&S->lsq, S->lsq.px, ssiz, h, tid);
            }
            __syncthreads();
            /* errs[0] = HDs(h); errs[4] = errs[0]; iterH */
            double itJ = 0; int itValid = 0;            /* iterH's maxS and whether H-out was updated */
            {
                double ths = DG_TC * th, dth = (ths - th) / DG_ILSQ_ITERS;
                dg_pass_cfg c1 = dg_cfg0(n); c1.wantJ = 1; c1.thJ = th; c1.list = intbuff; c1.thL = th;
                dg_pass_res r1 = dg_h_pass(c, h, c1); c.n_hds++;
                double mJ = r1.J; unsigned mI = r1.I;   /* maxS = inlidxs(errs[4], th) */
                /* hloc (= S->dtmp[0..8]) : the h being iterated; h (S->Hx) : iterH's output parameter */
                if (mI >= 4) {
                    double *hl = S->dtmp;
                    __syncthreads();
                    if (tid < 64) {
                        int cnt = (int)mI > (int)inlLimit ? (int)inlLimit : (int)mI;
                        { int id; if (mI > inlLimit) dg_randsubset_wave(&S->rng, intbuff, (int)mI, (int)inlLimit, tid, &id);
                            else id = tid < cnt ? intbuff[tid] : 0; dg_gather_wave(c, id, cnt, S->lsq.px, tid); }
                        DG_WSYNC();
                        dg_u2h_small_w(&S->lsq, S->lsq.px, cnt, hl, tid);
                    }
                    __syncthreads();
                    int early = 0;
                    for (int it = 0; it < DG_ILSQ_ITERS; ++it) {
                        dg_pass_cfg c2 = dg_cfg0(n); c2.wantJ = 1; c2.thJ = th; c2.list = intbuff; c2.thL = ths;
                        dg_pass_res r2 = dg_h_pass(c, hl, c2); c.n_hds++;
                        if (mJ < r2.J) { mJ = r2.J; mI = r2.I; __syncthreads(); if (tid < 9) h[tid] = hl[tid]; __syncthreads(); }
                        if (r2.nL < 4) { early = 1; break; }
                        __syncthreads();
                        if (tid < 64) {
                            int cnt = r2.nL > inlLimit ? (int)inlLimit : (int)r2.nL;
                            { int id; if (r2.nL > inlLimit) dg_randsubset_wave(&S->rng, intbuff, (int)r2.nL, (int)inlLimit, tid, &id);
                                else id = tid < cnt ? intbuff[tid] : 0; dg_gather_wave(c, id, cnt, S->lsq.px, tid); }
                            DG_WSYNC();
                            dg_u2h_small_w(&S->lsq, S->lsq.px, cnt, hl, tid);
                        }
                        __syncthreads();
                        ths -= dth;
                    }
                    if (!early) {
                        dg_pass_cfg c3 = dg_cfg0(n); c3.wantJ = 1; c3.thJ = th; c3.list = intbuff; c3.thL = th;
                        dg_pass_res r3 = dg_h_pass(c, hl, c3); c.n_hds++;
                        if (mJ < r3.J) { mJ = r3.J; mI = r3.I; __syncthreads(); if (tid < 9) h[tid] = hl[tid]; __syncthreads(); }
                    }
                    itJ = mJ; itValid = 1;
                } else { itJ = 0; itValid = 1; }         /* returns S = {0,0}: h untouched */
            }
            (void)itValid; (void)hbest;
            if (maxJ < itJ) { maxJ = itJ; __syncthreads(); if (tid < 9) H[tid] = h[tid]; __syncthreads(); }
        }
    }
    /* inl[j] = (errs[0][j] <= th) with errs[0] = residuals of the (possibly refined) H */
    dg_pass_cfg cf = dg_cfg0(n); cf.wantC = 1; cf.thC = th;
    {
        double Hr[9]; for (int i = 0; i < 9; i++) Hr[i] = H[i];
        const dg_pt *P = c.P;
        unsigned cnt = 0;
        for (int base = 0; base < n; base += DG_T) {
            int j = base + tid; bool in = false;
            if (j < n) { dg_pt p = dg_ldpt<LDSPTS>(P, j); in = dg_HDs(Hr, p.x1, p.y1, p.x2, p.y2) <= th; inl_flags[j] = in ? 1 : 0; }
            cnt += in ? 1u : 0u;
        }
        cnt = dg_block_sum_u(&c.S->red, cnt, tid);
        __syncthreads();
        return cnt;
    }
}

/* ---- innerH with one repetition per wave ----------------------------------------------------------------------------
 * The ten repetitions of inHrani (ranH.c:88-135) depend on each other through the generator, the order of `inliers` and the
 * running best only.  What a repetition draws is its sample (ssiz numbers) and one 10-subset per re-fit whose inlier list is
 * longer than inlLimit = 10 — five of them when the plane is real, which is the common case.  So a round of DG_NW repetitions
 * runs concurrently, one per wave, each on its own id list / MSAC-term buffer in the workspace (dg_f_cshared::wlist /
 * wstage) and its own solver scratch (dg_f_shared::ww): wave 0 first draws the round's samples one after the other from
 * generator states that ASSUME five subsets per earlier repetition of the round; afterwards the repetitions are committed
 * in order as long as that assumption held (the first one always does), the generator is set to the exact state behind
 * the last committed one, and the next round starts behind it (after putting `inliers` back into the order the last
 * committed sample left).  Every repetition performs the arithmetic of the serial order (same fits, same passes, J as the
 * reference's sequential sum), so results and counters are identical; dg_innerH_serial is kept behind
 * MI_DEGENSAC_TUNE_F_SERIAL_REPS for the equality test. */
/* One wave's pass over all n points (what dg_pass does for a workgroup): I = #(d <= thJ), J = the reference-order MSAC sum, the
 * ordered id lists at thL (la, when given) and thL2 (lb, when given).  The nonzero MSAC terms of a step (DG_PU tiles of 64 points)
 * go through `tile` (LDS, >= 64 * DG_PU doubles: the wave's solver scratch, idle during a pass) and are added, in point order,
 * before the next step, whose points are loaded before that. */
template <int LDSPTS, class Err>
__device__ __forceinline__ dg_pass_res dg_wpass_impl(const dg_pt *P, int n, Err err, double thJ, int *la_, double thL, int *lb_, double thL2, double *tile_,
    int lane)
{
    __attribute__((address_space(1))) int *la = (__attribute__((address_space(1))) int *)la_, *lb = (__attribute__((address_space(1))) int *)lb_;
    __attribute__((address_space(3))) double *t = (__attribute__((address_space(3))) double *)tile_;
    dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0; out.nL2 = 0; out.nJ = 0;
    const double t94 = thJ * 9 / 4;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned cI = 0, nJ = 0, nA = 0, nB = 0;
    double J = 0;
    dg_pt qn[DG_PU];
#pragma unroll
    for (int u = 0; u < DG_PU; u++) { const int j = u * 64 + lane; qn[u] = dg_ldpt<LDSPTS>(P, j < n ? j : 0); }
    DG_WSYNC();
    for (int base = 0; base < n; base += DG_PU * 64) {
        dg_pt q[DG_PU]; double d[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) q[u] = qn[u];
        if (LDSPTS != 1) {
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { const int j = base + DG_PU * 64 + u * 64 + lane; if (j < n) qn[u] = dg_ldpt<LDSPTS>(P, j); }
        }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) d[u] = err(q[u]);
        unsigned sJ = 0;
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            const int j = base + u * 64 + lane; const bool act = j < n;
            double term = 0.0;
            if (act && thJ != 0 && !(d[u] >= t94)) term = 1 - (d[u] / t94);
            const bool nz = !(term == 0.0), inA = la_ && act && d[u] <= thL, inB = lb_ && act && d[u] <= thL2;
            cI += (act && d[u] <= thJ) ? 1u : 0u;
            const unsigned long long mJ = __ballot(nz), mA = __ballot(inA), mB = __ballot(inB);
            if (nz) t[sJ + (unsigned)__popcll(mJ & below)] = term;
            if (inA) la[nA + (unsigned)__popcll(mA & below)] = j;
            if (inB) lb[nB + (unsigned)__popcll(mB & below)] = j;
            sJ += (unsigned)__popcll(mJ); nA += (unsigned)__popcll(mA); nB += (unsigned)__popcll(mB);
        }
        if (LDSPTS == 1) {
#pragma unroll
            for (int u = 0; u < DG_PU; u++) { const int j = base + DG_PU * 64 + u * 64 + lane; if (j < n) qn[u] = dg_ldpt<LDSPTS>(P, j); }
        }
        DG_WSYNC_LDS();                  /* (the terms are in LDS; the id lists, global, are read after the pass) */
        if (sJ) J = dg_seq_sum_impl<3>((const double *)tile_, (int)sJ, J);
        nJ += sJ;
        DG_WSYNC_LDS();
    }
    out.I = dg_wave_sum_u(cI); out.J = J; out.nL = nA; out.nL2 = nB; out.nJ = nJ;
    DG_WSYNC();
    return out;
}
/* One wave's share of a pass that a workgroup splits into point slices (dg_lo_rep_wg): the points [lo, hi), ids and MSAC terms
 * compacted in point order into slice-local outputs (la / lb / jout start at the slice's first slot); no sum: the caller adds the
 * slices' terms one after the other.  Returns I, nL, nL2, nJ of the slice. */
template <class Err>
__device__ __forceinline__ dg_pass_res dg_wpass_slice(const dg_pt *P, int lo, int hi, Err err, double thJ, int *la_, double thL, int *lb_, double thL2,
    double *jout_, int lane)
{
    __attribute__((address_space(1))) int *la = (__attribute__((address_space(1))) int *)la_, *lb = (__attribute__((address_space(1))) int *)lb_;
    __attribute__((address_space(1))) double *jo = (__attribute__((address_space(1))) double *)jout_;
    dg_pass_res out; out.I = 0; out.J = 0; out.C = 0; out.nL = 0; out.nF = 0; out.nL2 = 0; out.nJ = 0;
    const double t94 = thJ * 9 / 4;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned cI = 0, nJ = 0, nA = 0, nB = 0;
    dg_pt qn[DG_PU];
#pragma unroll
    for (int u = 0; u < DG_PU; u++) { const int j = lo + u * 64 + lane; qn[u] = dg_ldpt<0>(P, j < hi ? j : 0); }
    for (int base = lo; base < hi; base += DG_PU * 64) {
        dg_pt q[DG_PU]; double d[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) q[u] = qn[u];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { const int j = base + DG_PU * 64 + u * 64 + lane; if (j < hi) qn[u] = dg_ldpt<0>(P, j); }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) d[u] = err(q[u]);
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            const int j = base + u * 64 + lane; const bool act = j < hi;
            double term = 0.0;
            if (jout_ && act && !(d[u] >= t94)) term = 1 - (d[u] / t94);
            const bool nz = !(term == 0.0), inA = la_ && act && d[u] <= thL, inB = lb_ && act && d[u] <= thL2;
            cI += (act && d[u] <= thJ) ? 1u : 0u;
            const unsigned long long mJ = __ballot(nz), mA = __ballot(inA), mB = __ballot(inB);
            if (nz) jo[nJ + (unsigned)__popcll(mJ & below)] = term;
            if (inA) la[nA + (unsigned)__popcll(mA & below)] = j;
            if (inB) lb[nB + (unsigned)__popcll(mB & below)] = j;
            nJ += (unsigned)__popcll(mJ); nA += (unsigned)__popcll(mA); nB += (unsigned)__popcll(mB);
        }
    }
    out.I = dg_wave_sum_u(cI); out.nL = nA; out.nL2 = nB; out.nJ = nJ;
    return out;
}
static_assert(offsetof(dg_wave_ws, Z) == 0 && offsetof(dg_wave_ws, px) + sizeof(((dg_wave_ws *)0)->px) >= 64 * DG_PU * sizeof(double) &&
              offsetof(dg_wave_ws, ews) >= offsetof(dg_wave_ws, px) + sizeof(((dg_wave_ws *)0)->px),
                  "a wave pass's MSAC-term tile spans dg_wave_ws::Z .. ::px");

/* one wave's pass of homography Hm (metric HDs): I, J at thJ, the ordered id list at thL */
template <int LDSPTS>
__device__ __noinline__ dg_pass_res dg_hds_wpass(const dg_pt *P, int n, const double *Hm /* LDS */, double thJ, int *list_, double thL, double *tile, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n);
    double H[9];
#pragma unroll
    for (int i = 0; i < 9; i++) H[i] = Hm[i];
    return dg_wpass_impl<LDSPTS>(P, n, [&](const dg_pt &q) { return dg_HDs(H, q.x1, q.y1, q.x2, q.y2); }, thJ, list_, thL, (int *)0, 0.0, tile, lane);
}

/* one repetition of inHrani + iterH (ranH.c:88-135, :18-86) by one wave */
template <int LDSPTS>
__device__ __noinline__ void dg_innerH_rep_wave(CTX &c, dg_ih_log *lg, int ssiz, double th, int lim, int lane, int wave)
{
    dg_f_shared *S = c.S; const int n = c.n; const dg_pt *P = c.P;
    dg_wave_ws *w = &S->ww[wave];
    int *ib = c.K->wlist + (size_t)wave * c.K->n_max; double *jb = w->Z;                /* the pass's MSAC-term tile: Z .. px, idle during a pass */
    double *h = w->H, *hl = w->F;
    /* an earlier repetition of this round has finished with another number of draws than this one's start state assumes:
     * this repetition will not be committed, stop it */
    auto stale = [&]() {
        int bad = 0;
        if (lane < wave) { const int d = __hip_atomic_load(&S->ih[lane].pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); bad = d >= 0 && d != 5 * lim; }
        return __ballot(bad) != 0ull;
    };
    /* u2h on cnt >= 4 ids, lane j < cnt holding the j-th (Htools.c:101-133: 4 points exactly, else normalised) */
    auto fit = [&](int id, int cnt, double *dst) {
        DG_WSYNC();
        dg_gather_wave(c, id, cnt, w->px, lane);
        DG_WSYNC();
        if (cnt == 4) { if (lane == 0) dg_u2h_4pt_mv(w->Z, w->V, w->px, dst); DG_WSYNC(); }
        else dg_u2h_norm_wave_noz(w, w->px, cnt, dst, lane);
    };
    /* the ids of a list of `len` (> 4) entries the next fit uses: a random 10-subset when it is longer than inlLimit */
    auto pick = [&](int len, int *cnt, int *draws) {
        int id = 0;
        *cnt = len > lim ? lim : len;
        if (len > lim) { dg_randsubset_wave(&lg->g, ib, len, lim, lane, &id); *draws += lim; }
        else id = lane < len ? ib[lane] : 0;
        return id;
    };
    int passes = 0, draws = 0;
    fit(lane < ssiz ? lg->ids[lane] : 0, ssiz, h);
    const dg_pass_res r1 = dg_hds_wpass<LDSPTS>(P, n, h, th, ib, th, jb, lane); passes++;
    double mJ = r1.J, itJ = 0; unsigned mI = r1.I;
    if (mI >= 4) {
        double ths = DG_TC * th; const double dth = (ths - th) / DG_ILSQ_ITERS;
        { int cnt; const int id = pick((int)mI, &cnt, &draws); fit(id, cnt, hl); }
        int early = 0;
        for (int it = 0; it < DG_ILSQ_ITERS; ++it) {
            if (stale()) { if (lane == 0) lg->aborted = 1; DG_WSYNC(); return; }
            const dg_pass_res r2 = dg_hds_wpass<LDSPTS>(P, n, hl, th, ib, ths, jb, lane); passes++;
            if (mJ < r2.J) { mJ = r2.J; mI = r2.I; DG_WSYNC(); if (lane < 9) h[lane] = hl[lane]; DG_WSYNC(); }
            if (r2.nL < 4) { early = 1; break; }
            { int cnt; const int id = pick((int)r2.nL, &cnt, &draws); fit(id, cnt, hl); }
            ths -= dth;
        }
        if (!early) {
            const dg_pass_res r3 = dg_hds_wpass<LDSPTS>(P, n, hl, th, ib, th, jb, lane); passes++;
            if (mJ < r3.J) { mJ = r3.J; mI = r3.I; DG_WSYNC(); if (lane < 9) h[lane] = hl[lane]; DG_WSYNC(); }
        }
        itJ = mJ;
    }
    DG_WSYNC();
    if (lane < 9) lg->h[lane] = h[lane];
    if (lane == 0) { lg->itJ = itJ; lg->passes = passes; lg->draws = draws; __hip_atomic_store(&lg->pub, draws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    DG_WSYNC();
}

template <int LDSPTS>
__device__ __noinline__ unsigned dg_innerH_waves(CTX &c, double *H /* LDS, in/out */, double th, unsigned inlLimit, unsigned char *inl_flags)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid, lane = tid & 63, wave = tid >> 6;
    int *inliers = c.K->L[3], *bak = c.K->L[4];
    const int lim = (int)inlLimit;
    dg_pass_cfg cfg = dg_cfg0(n); cfg.list = inliers; cfg.thL = th;
    dg_pass_res r0 = dg_h_pass(c, H, cfg); c.n_hds++;
    const int ninl = (int)r0.nL;
    if (ninl >= 8) {
        int ssiz = ninl / 2; if (ssiz > 12) ssiz = 12;
        double maxJ = 0;
        int next = 0;
        while (next < DG_RAN_REP) {
            const int nr = DG_RAN_REP - next < DG_NW ? DG_RAN_REP - next : DG_NW;
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(wave) == 0) {
                /* the round's samples, drawn one after the other from ASSUMED generator states (5 subsets of `lim` per repetition) */
                if (lane == 0) { S->ih_start = S->rng; S->ih_work = S->rng; }
                for (int j = lane; j < ninl; j += 64) bak[j] = inliers[j];
                DG_WSYNC();
                for (int q = 0; q < nr; q++) {
                    int id = 0;
                    dg_randsubset_wave(&S->ih_work, inliers, ninl, ssiz, lane, &id);
                    if (lane < ssiz) S->ih[q].ids[lane] = id;
                    if (lane == 0) { S->ih[q].g = S->ih_work; S->ih[q].pub = -1; S->ih[q].aborted = 0; }
                    DG_WSYNC();
                    dg_rand_skip(&S->ih_work, 5 * (int)lim, lane);
                }
            }
            __syncthreads();
            if (wave < nr) dg_innerH_rep_wave<LDSPTS>(c, &S->ih[wave], ssiz, th, lim, lane, wave);
            __syncthreads();
            /* commit in order while the assumption behind each repetition's start state held */
            int v = 0;
            for (int q = 0; q < nr; q++) {
                if (S->ih[q].aborted) break;                     /* stopped as stale: it runs again in the next round (q >= 1 here) */
                c.n_hds += S->ih[q].passes;
                if (maxJ < S->ih[q].itJ) { maxJ = S->ih[q].itJ; __syncthreads(); if (tid < 9) H[tid] = S->ih[q].h[tid]; __syncthreads(); }
                v++;
                if (S->ih[q].draws != 5 * lim) break;
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(wave) == 0) {
                if (lane == 0) S->rng = S->ih[v - 1].g;               /* the exact state behind repetition next + v - 1 */
                if (v < nr) {
                    /* `inliers` as that repetition's sample left it: back to the round's start, the committed samples again */
                    for (int j = lane; j < ninl; j += 64) inliers[j] = bak[j];
                    if (lane == 0) S->ih_work = S->ih_start;
                    DG_WSYNC();
                    for (int q = 0; q < v; q++) {
                        int id = 0;
                        dg_randsubset_wave(&S->ih_work, inliers, ninl, ssiz, lane, &id);
                        dg_rand_skip(&S->ih_work, S->ih[q].draws, lane);
                    }
                }
            }
            next += v;
        }
        __syncthreads();
    }
    /* inl[j] = (errs[0][j] <= th) with errs[0] = residuals of the (possibly refined) H */
    {
        double Hr[9]; for (int i = 0; i < 9; i++) Hr[i] = H[i];
        const dg_pt *P = c.P;
        unsigned cnt = 0;
        for (int base = 0; base < n; base += DG_T) {
            int j = base + tid; bool in = false;
            if (j < n) { dg_pt p = dg_ldpt<LDSPTS>(P, j); in = dg_HDs(Hr, p.x1, p.y1, p.x2, p.y2) <= th; inl_flags[j] = in ? 1 : 0; }
            cnt += in ? 1u : 0u;
        }
        cnt = dg_block_sum_u(&c.S->red, cnt, tid);
        __syncthreads();
        return cnt;
    }
}

template <int LDSPTS>
__device__ __forceinline__ unsigned dg_innerH(CTX &c, double *H /* LDS, in/out */, double th, unsigned inlLimit, unsigned char *inl_flags)
{
    /* one repetition per wave needs inlLimit <= 14 gathered points per fit (dg_wave_ws::px); the driver passes 10 (DegUtils.c:693-731) */
    if (c.A->innerh_serial || inlLimit > 12) return dg_innerH_serial<LDSPTS>(c, H, th, inlLimit, inl_flags);
    return dg_innerH_waves<LDSPTS>(c, H, th, inlLimit, inl_flags);
}

/* ---- wave-level passes (one wave, no workgroup barriers) ------------------------------------------ */
/* #points with Sampson error < thr (strict, DegUtils.c style); DG_PU points per lane and step (loads in flight together) */
template <int LDSPTS>
__device__ __forceinline__ unsigned dg_wave_count_lt(const dg_pt *P, int n, const double *F, double thr, int lane)
{
    unsigned cnt = 0;
    for (int p0 = lane; p0 < n; p0 += 64 * DG_PU) {
        dg_pt q[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { const int p = p0 + 64 * u; q[u] = dg_ldpt<LDSPTS>(P, p < n ? p : p0); }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) cnt += (p0 + 64 * u < n && dg_FDs(F, q[u].x1, q[u].y1, q[u].x2, q[u].y2) < thr) ? 1u : 0u;
    }
    return dg_wave_sum_u(cnt);
}
/* ordered id list of the points with error < thr; returns the count */
template <int LDSPTS>
__device__ __forceinline__ unsigned dg_wave_list_lt(const dg_pt *P, int n, const double *F, double thr, int *list, int lane)
{
    unsigned off = 0;
    for (int base = 0; base < n; base += 64 * DG_PU) {
        dg_pt q[DG_PU]; bool in[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * u + lane; q[u] = dg_ldpt<LDSPTS>(P, p < n ? p : 0); }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) in[u] = base + 64 * u + lane < n && dg_FDs(F, q[u].x1, q[u].y1, q[u].x2, q[u].y2) < thr;
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            const unsigned long long b = __ballot(in[u]);
            if (in[u]) list[off + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = base + 64 * u + lane;
            off += (unsigned)__popcll(b);
        }
    }
    return off;
}

/* dg_wave_list_lt that also leaves the listed points themselves, in list order, in `stage` (global): stage[j] = P[list[j]].  The pass has every
 * listed point in registers when it decides; gathering them again through the finished list costs an id load and a point load per 64 ids, one
 * dependent pair of round trips after the other (round 6: 13 such steps per u2Fit iteration on 800 ids). */
template <int LDSPTS>
__device__ __forceinline__ unsigned dg_wave_list_lt_stage(const dg_pt *P, int n, const double *F, double thr, int *list, dg_pt *stage_, int lane)
{
    typedef __attribute__((address_space(1))) double dg_gdbl;
    dg_gdbl *stage = (dg_gdbl *)(double *)stage_;
    unsigned off = 0;
    for (int base = 0; base < n; base += 64 * DG_PU) {
        dg_pt q[DG_PU]; bool in[DG_PU];
#pragma unroll
        for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * u + lane; q[u] = dg_ldpt<LDSPTS>(P, p < n ? p : 0); }
#pragma unroll
        for (int u = 0; u < DG_PU; u++) in[u] = base + 64 * u + lane < n && dg_FDs(F, q[u].x1, q[u].y1, q[u].x2, q[u].y2) < thr;
#pragma unroll
        for (int u = 0; u < DG_PU; u++) {
            const unsigned long long b = __ballot(in[u]);
            if (in[u]) {
                const unsigned pos = off + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
                list[pos] = base + 64 * u + lane;
                dg_gdbl *o = stage + 4 * (size_t)pos; o[0] = q[u].x1; o[1] = q[u].y1; o[2] = q[u].x2; o[3] = q[u].y2;
            }
            off += (unsigned)__popcll(b);
        }
    }
    return off;
}

/* ---- DegUtils.c:635-690 u2Fit, run by ONE wave on its own scratch -----------------------------------
 * F (LDS, 9) in/out.  Returns the count and *thf = the threshold the reference's `inl` flags correspond to
 * (th after the full schedule, the current ths on the "fewer than 8 inliers" early return). */
template <int LDSPTS>
__device__ __noinline__ unsigned dg_u2Fit_wave(dg_wave_ws *w, const dg_pt *P, int n, double *F, double th, double ths, unsigned iters,
                                              int *list, dg_pt *stage, int stage_cap, int lane, double *thf, int *n_aux)
{
    double dth = (ths - th) / (iters - 1);
    for (unsigned iter = 0; iter < iters; ++iter) {
        double Fr[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Fr[i] = F[i];
        unsigned cnt = dg_wave_list_lt_stage<LDSPTS>(P, n, Fr, ths, list, stage, lane); (*n_aux)++;
        if (cnt < 8) { *thf = ths; return cnt; }
        DG_WSYNC();
        if (cnt <= 14) {
            if (lane < (int)cnt) { dg_pt q = dg_ldpt<LDSPTS>(P, list[lane]); w->px[4*lane] = q.x1; w->px[4*lane+1] = q.y1; w->px[4*lane+2] = q.x2;
                w->px[4*lane+3] = q.y2; }
            DG_WSYNC();
            if (cnt > 8) dg_u2f_norm_w(w, w->px, (const double *)0, (int)cnt, F, lane);
            else {
                /* exactly 8: the svduv path of u2f (Ftools.c:371-384) */
                if (lane == 0) { for (int i = 0; i < 72; i++) w->Z[i] = 0.;
                    for (int i = 0; i < 8; i++) { double a[3] = {w->px[4*i], w->px[4*i+1], 1.0}, b[3] = {w->px[4*i+2], w->px[4*i+3], 1.0};
                    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) w->Z[(k*3+l)*8 + i] = b[k] * a[l]; } }
                DG_WSYNC();
                dg_svd_lastcol_9x8_wave(w->Z, w->V, lane);
                if (lane == 0) { for (int i = 0; i < 9; i++) F[i] = w->V[i]; dg_singulF(F); }
                DG_WSYNC();
            }
        } else {
            /* (the listed points are in `stage` already: dg_wave_list_lt_stage; the DG_WSYNC above has drained its stores) */
            /* the normal matrix from shared design-matrix entries, twelve points per fill of this wave's Z (idle in the long form) */
            /* (with an LDS table the fit needs no scratch behind the staged points: any list length) */
            /* the table: Z and V of this wave's scratch, 200 doubles (V is written when the sums are done) */
            static_assert(offsetof(dg_wave_ws, V) == offsetof(dg_wave_ws, Z) + sizeof(((dg_wave_ws *)0)->Z) && sizeof(((dg_wave_ws *)0)->Z) +
                          sizeof(((dg_wave_ws *)0)->V) >= 200 * sizeof(double), "table of the long-list fit");
            dg_lsq_seq_par(w, stage, (int)cnt, lane, 64, 0, w->A1, w->A2, [] { DG_WSYNC(); }, w->Z, 20);
            DG_WSYNC();
            dg_eig_sym_wave(w->V, w->D, lane, &w->ews);
            if (lane == 0) {
                int jm = 0; for (int i = 1; i < 9; i++) if (w->D[i] < w->D[jm]) jm = i;
                for (int i = 0; i < 9; i++) F[i] = w->V[jm*9 + i];
                dg_singulF(F);
                dg_denormF(F, w->A1, w->A2);
            }
            DG_WSYNC();
        }
        ths -= dth;
    }
    double Fr[9];
#pragma unroll
    for (int i = 0; i < 9; i++) Fr[i] = F[i];
    unsigned cnt = dg_wave_count_lt<LDSPTS>(P, n, Fr, th, lane); (*n_aux)++;
    *thf = th;
    return cnt;
}

/* ---- DegUtils.c:488-632 innerFH + dual_sample ---------------------------------------------------- */
/* dual_sample draws on freshly initialised identity permutations; only the first sA (sB) entries are
 * read afterwards, so the permutation is tracked sparsely on lane 0. */
__device__ __forceinline__ void dg_dual_pick(const int *raw /* s rand() outputs, in draw order */, unsigned len, unsigned s, int *out /* s entries */)
{
    int pos_key[16], pos_val[16], np = 0;
    for (unsigned pos = 0; pos < s; ++pos) {
        unsigned idx = (unsigned)raw[pos] % len;
        /* swap ptr[pos] <-> ptr[idx] on a sparse identity map */
        int vp = (int)pos, vi = (int)idx, ip = -1, ii = -1;
        for (int k = 0; k < np; k++) { if (pos_key[k] == (int)pos) { vp = pos_val[k]; ip = k; } if (pos_key[k] == (int)idx) { vi = pos_val[k]; ii = k; } }
        if (ip < 0) { ip = np; pos_key[np++] = (int)pos; }
        pos_val[ip] = vi;
        if ((int)idx != (int)pos) { if (ii < 0) { ii = np; pos_key[np++] = (int)idx; } pos_val[ii] = vp; }
        else pos_val[ip] = vp;
    }
    for (unsigned i = 0; i < s; i++) {
        int v = (int)i;
        for (int k = 0; k < np; k++) if (pos_key[k] == (int)i) v = pos_val[k];
        out[i] = v;
    }
}

/* The repetitions of innerFH are independent of each other: dual_sample always consumes 6 + 4 draws, u2f / u2Fit
 * consume none, and whether a repetition is refined by u2Fit depends only on the running maximum of the
 * pre-refinement counts.  So: lane 0 draws all samples; the waves then fit and score the repetitions in
 * parallel; the "new record" repetitions are refined in parallel (one wave each); a final scan in repetition
 * order applies the reference's `max_i < no_i` bookkeeping and one pass materialises the winner's flags. */
template <int LDSPTS>
__device__ __noinline__ void dg_innerFH(CTX &c, const int *idxH, unsigned lenH, const int *idxO, unsigned lenO,
                                           double th, unsigned repCount, double *F /* LDS out */, unsigned char *inl /* out flags */)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid, lane = tid & 63, wave = tid >> 6;
    const dg_pt *P = c.P;
    __syncthreads();
    /* dual_sample consumes exactly 6 + 4 rand() outputs per repetition, whatever they are: lane 0 draws them all,
     * then one lane per repetition replays its two sparse permutations */
    if (__builtin_amdgcn_readfirstlane(wave) == 0)
        for (int q0 = 0; q0 < (int)repCount * 10; q0 += 30) {              /* 30 draws = three repetitions per block */
            const int m = (int)repCount * 10 - q0 < 30 ? (int)repCount * 10 - q0 : 30;
            const int v = dg_rand_block(&S->rng, m, lane);
            if (lane < m) S->fhRaw[q0 + lane] = v;
        }
    __syncthreads();
    if (tid < (int)repCount) {
        const unsigned rep = (unsigned)tid;
        int pick[16];
        dg_dual_pick(S->fhRaw + rep * 10, lenH, 6, pick);
        dg_dual_pick(S->fhRaw + rep * 10 + 6, lenO, 4, pick + 6);
        for (int i = 0; i < 6; i++) S->fhIds[rep][i] = idxH[pick[i]];
        for (int i = 0; i < 4; i++) S->fhIds[rep][6+i] = idxO[pick[6+i]];
    }
    __syncthreads();
#if DG_NW >= 8
    /* 10-point model + its consensus, one wave per repetition */
    for (unsigned rep = wave; rep < repCount; rep += DG_NW) {
        dg_wave_ws *w = &S->ww[wave];
        if (lane < 10) { dg_pt q = dg_ldpt<LDSPTS>(P, S->fhIds[rep][lane]); w->px[4*lane] = q.x1; w->px[4*lane+1] = q.y1; w->px[4*lane+2] = q.x2;
            w->px[4*lane+3] = q.y2; }
        DG_WSYNC();
        dg_u2f_norm_w(w, w->px, (const double *)0, 10, S->fhF[rep], lane);
        double Fr[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Fr[i] = S->fhF[rep][i];
        unsigned cnt = dg_wave_count_lt<LDSPTS>(P, n, Fr, th, lane);
        if (lane == 0) { S->fhCnt[rep] = (int)cnt; S->fhCnt2[rep] = -1; }
        DG_WSYNC();
    }
#else
    /* 10-point model + its consensus: a wave takes two repetitions at a time, one fit per half-wave (dg_fit_norm_w2), then their two
     * counting passes.  Fifteen repetitions on four waves: two rounds of fits instead of four. */
    static_assert(sizeof(dg_lsq_scratch) >= (size_t)DG_NW * DG_X2_DOUBLES * sizeof(double), "second-problem scratch");
    for (unsigned base = wave; base < repCount; base += 2 * DG_NW) {
        dg_wave_ws *w = &S->ww[wave];
        double *x2 = (double *)&S->lsq + (size_t)wave * DG_X2_DOUBLES, *pxB = w->Z + 80;
        const unsigned repB = base + DG_NW; const bool two = repB < repCount;
        if (lane < 10) { dg_pt q = dg_ldpt<LDSPTS>(P, S->fhIds[base][lane]); w->px[4*lane] = q.x1; w->px[4*lane+1] = q.y1; w->px[4*lane+2] = q.x2;
            w->px[4*lane+3] = q.y2; }
        else if (two && lane >= 32 && lane < 42) { const int hl = lane - 32; dg_pt q = dg_ldpt<LDSPTS>(P, S->fhIds[repB][hl]); pxB[4*hl] = q.x1;
            pxB[4*hl+1] = q.y1; pxB[4*hl+2] = q.x2; pxB[4*hl+3] = q.y2; }
        DG_WSYNC();
        dg_fit_norm_w2<false>(w, x2, w->px, pxB, 10, S->fhF[base], two ? S->fhF[repB] : (double *)0, lane);
        for (int h = 0; h < (two ? 2 : 1); h++) {
            const unsigned rep = h ? repB : base;
            double Fr[9];
#pragma unroll
            for (int i = 0; i < 9; i++) Fr[i] = S->fhF[rep][i];
            unsigned cnt = dg_wave_count_lt<LDSPTS>(P, n, Fr, th, lane);
            if (lane == 0) { S->fhCnt[rep] = (int)cnt; S->fhCnt2[rep] = -1; }
            DG_WSYNC();
        }
    }
#endif
    c.n_aux += (int)repCount;
    __syncthreads();
    /* repetitions that set a new record of the pre-refinement count get u2Fit (DegUtils.c:562-566) */
    int nfit = 0, fitrep[16];
    { unsigned max_s = 0; for (unsigned rep = 0; rep < repCount; ++rep) if ((unsigned)S->fhCnt[rep] > max_s) { max_s = (unsigned)S->fhCnt[rep];
        fitrep[nfit++] = (int)rep; } }
    int aux_local = 0;
    for (int q = wave; q < nfit; q += DG_NW) {
        const int rep = fitrep[q];
        if (lane < 9) S->fhF2[rep][lane] = S->fhF[rep][lane];
        DG_WSYNC();
        double thf;
        unsigned cnt = dg_u2Fit_wave<LDSPTS>(&S->ww[wave], P, n, S->fhF2[rep], th, th*3, 4, c.K->wlist + (size_t)wave * c.K->n_max,
            c.K->wstage + (size_t)wave * c.K->n_max, c.K->n_max, lane, &thf, &aux_local);
        if (lane == 0) { S->fhCnt2[rep] = (int)cnt; S->fhTh[rep] = thf; S->itmp[8 + wave] = aux_local; }
        DG_WSYNC();
    }
    if (lane == 0 && wave >= nfit) S->itmp[8 + wave] = 0;
    if (lane == 0 && wave < nfit) S->itmp[8 + wave] = aux_local;
    __syncthreads();
    for (int w = 0; w < DG_NW; w++) c.n_aux += S->itmp[8 + w];
    /* the reference's bookkeeping, in repetition order */
    unsigned max_i = 0; int best = -1, best_fit = 0;
    for (unsigned rep = 0; rep < repCount; ++rep) {
        if (max_i < (unsigned)S->fhCnt[rep]) { max_i = (unsigned)S->fhCnt[rep]; best = (int)rep; best_fit = 0; }
        if (S->fhCnt2[rep] >= 0 && max_i < (unsigned)S->fhCnt2[rep]) { max_i = (unsigned)S->fhCnt2[rep]; best = (int)rep; best_fit = 1; }
    }
    __syncthreads();
    if (best < 0) {
        if (tid < 9) F[tid] = 1;
        for (int j = tid; j < n; j += DG_T) inl[j] = 0;
    } else {
        const double *Fb = best_fit ? S->fhF2[best] : S->fhF[best];
        const double thb = best_fit ? S->fhTh[best] : th;
        double Fr[9];
#pragma unroll
        for (int i = 0; i < 9; i++) Fr[i] = Fb[i];
        if (tid < 9) F[tid] = Fr[tid];
        for (int j = tid; j < n; j += DG_T) { dg_pt q = dg_ldpt<LDSPTS>(P, j); inl[j] = dg_FDs(Fr, q.x1, q.y1, q.x2, q.y2) < thb ? 1 : 0; }
    }
    __syncthreads();
}

/* ---- DegUtils.c:254-444 rFtH: plane-and-parallax completion -------------------------------------
 * The 2-point epipole hypotheses are evaluated in batches of DG_CHUNK: lane 0 replays the draws and the
 * ptr[] swaps for the whole batch, one wave scores each candidate on the off-plane points, and the first
 * candidate that beats m_i (the only kind that has side effects) is then processed exactly as the
 * sequential loop would: RNG and ptr[] are rolled back to their state right after that iteration. */
template <int LDSPTS>
__device__ __forceinline__ void dg_rFtH_aFt(const double *Hr, const dg_pt &p0, const dg_pt &p1, double *aFt)
{
    double a0[3] = {p0.x1, p0.y1, 1.0}, a1[3] = {p1.x1, p1.y1, 1.0}, b0[3], b1[3], c1[3], c2[3], ec[3];
    b0[0] = Hr[0]*p0.x2 + Hr[3]*p0.y2 + Hr[6]*1.0; b0[1] = Hr[1]*p0.x2 + Hr[4]*p0.y2 + Hr[7]*1.0; b0[2] = Hr[2]*p0.x2 + Hr[5]*p0.y2 + Hr[8]*1.0;
    b1[0] = Hr[0]*p1.x2 + Hr[3]*p1.y2 + Hr[6]*1.0; b1[1] = Hr[1]*p1.x2 + Hr[4]*p1.y2 + Hr[7]*1.0; b1[2] = Hr[2]*p1.x2 + Hr[5]*p1.y2 + Hr[8]*1.0;
    c1[0] = a0[1]*b0[2] - a0[2]*b0[1]; c1[1] = a0[2]*b0[0] - a0[0]*b0[2]; c1[2] = a0[0]*b0[1] - a0[1]*b0[0];
    c2[0] = a1[1]*b1[2] - a1[2]*b1[1]; c2[1] = a1[2]*b1[0] - a1[0]*b1[2]; c2[2] = a1[0]*b1[1] - a1[1]*b1[0];
    ec[0] = c1[1]*c2[2] - c1[2]*c2[1]; ec[1] = c1[2]*c2[0] - c1[0]*c2[2]; ec[2] = c1[0]*c2[1] - c1[1]*c2[0];
    double ecNorm = sqrt(ec[0]*ec[0] + ec[1]*ec[1] + ec[2]*ec[2]);
    ec[0] = ec[0]/ecNorm; ec[1] = ec[1]/ecNorm; ec[2] = ec[2]/ecNorm;
    /* aFt = ([e]x H^T)^T, every entry a three-term sum over the inner index in ascending order from 0.0 (DegUtils.c:337-345) */
    const double sk[9] = {0, -ec[2], ec[1], ec[2], 0, -ec[0], -ec[1], ec[0], 0};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { double z = 0.; z += sk[3*i] * Hr[3*j]; z += sk[3*i + 1] * Hr[3*j + 1]; z += sk[3*i + 2] * Hr[3*j + 2]; aFt[3*j + i] = z; }
}

template <int LDSPTS>
__device__ __noinline__ unsigned dg_rFtH(CTX &c, const unsigned char *hinl, double th, const double *H /* LDS */, double *F /* LDS out */)
{
    dg_f_shared *S = c.S; const int n = c.n, tid = c.tid, lane = tid & 63, wave = tid >> 6;
    unsigned char *nhinl = c.K->Fl[1], *vN = c.K->Fl[2], *inl = c.K->Fl[4];
    int *idxN = c.K->L[5], *idxH = c.K->L[6], *idxV = c.K->L[7];
    const unsigned MAX_SAM = 10000; const double conf = .999;
    double Hr[9]; for (int i = 0; i < 9; i++) Hr[i] = H[i];
    const dg_pt *P = c.P;
    unsigned nhinlCount, hinlCount;
    {
        dg_pass_cfg cfg = dg_cfg0(n); cfg.list = idxN; cfg.thL = 0.5; cfg.flags = nhinl; cfg.thF = 0.5;
        dg_pass_res r = dg_pass(&S->red, cfg, [&](int pid, int) { dg_pt p = dg_ldpt<LDSPTS>(P, pid); return dg_HDs(Hr, p.x1, p.y1, p.x2,
            p.y2) > 100*th ? 0.0 : 1.0; }, tid);
        c.n_hds++;
        nhinlCount = r.nL;
        dg_pass_cfg cfg2 = dg_cfg0(n); cfg2.list = idxH; cfg2.thL = 0.5;
        dg_pass_res r2 = dg_pass(&S->red, cfg2, [&](int pid, int) { return hinl[pid] ? 0.0 : 1.0; }, tid);
        hinlCount = r2.nL;
    }
    if (nhinlCount < 4 || hinlCount < 6) return 0;
    /* ptr[] lives in the sampler pool's LDS for the duration (the pool is parked in global memory) */
    int *ptr = LDSPTS != 0 ? c.pool : c.K->L[8];
    if (LDSPTS != 0) { for (int j = tid; j < n; j += DG_T) c.K->L[8][j] = c.pool[j]; __syncthreads(); }
    for (int j = tid; j < (int)nhinlCount; j += DG_T) ptr[j] = j;
    __syncthreads();
    unsigned max_i = 3, m_i = 4, max_sam = MAX_SAM;
    unsigned no_sam = 1;
    /* batch size: the candidates behind the first one that beats m_i are thrown away (the loop restarts behind it), and records are
     * front-loaded — with m_i = 4 nearly every candidate beats it, later ones rarely do.  So batches start small and double while
     * they bring no hit: the first hits cost a handful of scored candidates each instead of a full batch */
    int Bcap = 2 * DG_NW < DG_CHUNK ? 2 * DG_NW : DG_CHUNK;       /* measured on C2 x 4096: 2 .. 16 equal within the noise, 32 and more slower */
    if (c.A->dev_knob > 0) { Bcap = c.A->dev_knob & 0xffff; if (Bcap > DG_CHUNK) Bcap = DG_CHUNK; }
    while (no_sam < 2*max_sam) {
        int B = (int)(2*max_sam - no_sam); if (B > Bcap) B = Bcap;
        __syncthreads();
        long long tg0 = DG_CLK();
        if (tid == 0) {
            S->rng_save = S->rng;
            for (int b = 0; b < B; b++) {
                for (unsigned pos = 0; pos < 2; ++pos) {
                    unsigned idx = pos + 1 + (unsigned)dg_rand(&S->rng) % (nhinlCount - pos - 1);
                    int aux = ptr[pos]; ptr[pos] = ptr[idx]; ptr[idx] = aux;
                    c.K->rf[b][2 + pos] = (int)idx;
                }
                c.K->rf[b][0] = idxN[ptr[0]]; c.K->rf[b][1] = idxN[ptr[1]];
            }
        }
        __syncthreads();
        long long tg1 = DG_CLK(); DG_DEVT(if (tid == 0) S->dbg[1] += tg1 - tg0);
        /* one wave per candidate: #off-plane points with Sampson error < 2 th */
        for (int b = wave; b < B; b += DG_NW) {
            double aFt[9];
            dg_rFtH_aFt<LDSPTS>(Hr, dg_ldpt<LDSPTS>(P, c.K->rf[b][0]), dg_ldpt<LDSPTS>(P, c.K->rf[b][1]), aFt);
            unsigned cnt = 0;
            for (int j0 = lane; j0 < (int)nhinlCount; j0 += 64 * DG_PU) {     /* ids, then points, of DG_PU tiles in flight together */
                int id[DG_PU]; dg_pt q[DG_PU];
#pragma unroll
                for (int u = 0; u < DG_PU; u++) { const int j = j0 + 64 * u; id[u] = idxN[j < (int)nhinlCount ? j : j0]; }
#pragma unroll
                for (int u = 0; u < DG_PU; u++) q[u] = dg_ldpt<LDSPTS>(P, id[u]);
#pragma unroll
                for (int u = 0; u < DG_PU; u++) cnt += (j0 + 64 * u < (int)nhinlCount && dg_FDs(aFt, q[u].x1, q[u].y1, q[u].x2, q[u].y2) < th*2) ? 1u : 0u;
            }
            cnt = dg_wave_sum_u(cnt);
            if (lane == 0) c.K->rf[b][4] = (int)cnt;
        }
        __syncthreads();
        long long tg2 = DG_CLK(); DG_DEVT(if (tid == 0) S->dbg[2] += tg2 - tg1);
        /* first candidate beating m_i */
        bool hit = tid < B && (unsigned)c.K->rf[tid][4] > m_i;
        unsigned long long bal = __ballot(hit);
        __syncthreads();
        if (lane == 0) S->wave_cnt[wave] = bal ? (unsigned)(wave * 64 + __ffsll((long long)bal) - 1) : 0xffffffffu;
        __syncthreads();
        unsigned bE = S->wave_cnt[0];
        for (int w = 1; w < DG_NW; w++) bE = S->wave_cnt[w] < bE ? S->wave_cnt[w] : bE;
        if (bE == 0xffffffffu) { no_sam += (unsigned)B; c.n_aux += B; Bcap = 2 * Bcap < DG_CHUNK ? 2 * Bcap : DG_CHUNK; continue; }
        /* roll back to the state right after iteration bE */
        __syncthreads();
        if (tid == 0) {
            for (int b = B - 1; b > (int)bE; b--)
                for (int pos = 1; pos >= 0; --pos) { int idx = c.K->rf[b][2 + pos]; int aux = ptr[pos]; ptr[pos] = ptr[idx]; ptr[idx] = aux; }
            S->rng = S->rng_save;
            for (int q = 0; q < 2 * ((int)bE + 1); q++) dg_rand(&S->rng);
        }
        __syncthreads();
        no_sam += bE + 1; c.n_aux += (int)bE + 1;
        {
            double aFt[9];
            dg_rFtH_aFt<LDSPTS>(Hr, dg_ldpt<LDSPTS>(P, c.K->rf[bE][0]), dg_ldpt<LDSPTS>(P, c.K->rf[bE][1]), aFt);
            /* v = Ds < 2 th ; uV = uN(:, v) */
            dg_pass_cfg cfg = dg_cfg0((int)nhinlCount); cfg.src = idxN; cfg.flags = vN; cfg.thF = th*2;
            dg_pass(&S->red, cfg, [&](int pid, int) { dg_pt p = dg_ldpt<LDSPTS>(P, pid); return dg_FDs(aFt, p.x1, p.y1, p.x2, p.y2); }, tid);
            const unsigned char *vf = vN;
            dg_pass_cfg cf2 = dg_cfg0((int)nhinlCount); cf2.src = idxN; cf2.list = idxV; cf2.thL = 0.5;
            dg_pass_res rv = dg_pass(&S->red, cf2, [&](int, int pos) { return vf[pos] ? 0.0 : 1.0; }, tid);
            unsigned no_i = rv.nL;
            m_i = no_i;
            /* innerFH needs the sampler pool's LDS untouched?  It does not use ptr/pool; safe. */
            dg_innerFH(c, idxH, hinlCount, idxV, no_i, th, 15, S->ftmp, inl);
            unsigned ninl = 0, maxni = 0;
            for (int j = tid; j < n; j += DG_T) { if (inl[j]) { ninl++; if (nhinl[j]) maxni++; } }
            ninl = dg_block_sum_u(&S->red, ninl, tid);
            maxni = dg_block_sum_u(&S->red, maxni, tid);
            if (ninl > max_i) {
                max_i = ninl;
                __syncthreads();
                if (tid < 9) F[tid] = S->ftmp[tid];
                __syncthreads();
                unsigned ns = (unsigned)dg_nsamples((int)maxni, (int)nhinlCount, 2, conf);
                max_sam = max_sam > ns ? ns : max_sam;
            }
        }
        DG_DEVT(if (tid == 0) S->dbg[3] += DG_CLK() - tg2);
    }
    __syncthreads();
    if (LDSPTS != 0) { for (int j = tid; j < n; j += DG_T) c.pool[j] = c.K->L[8][j]; __syncthreads(); }
    return max_i;
}

#endif /* DG_KERNEL_F_H */
