/* mi_degensac — C-ABI of the MI355X-native LO-RANSAC / DEGENSAC estimator (libmi_degensac.so).
 *
 * Drop-in boundary for the reference's pybind11 FFI (src/pydegensac/bindings.cpp):
 *   findFundamentalMatrix_  bindings.cpp:253-467  -> mi_degensac_find_fundamental[_batch|_batch_dev]
 *   findHomography_         bindings.cpp:19-251   -> mi_degensac_find_homography[_batch|_batch_dev]
 * which in turn replace the two C drivers those bindings call:
 *   exp_ransacFcustomLAF    degensac/exp_ranF.h:72-74 (exp_ranF.c:1244-1767)
 *   exp_ransacHcustomLAF    degensac/exp_ranH.h:27-33 (exp_ranH.c:470-930)
 *
 * Plain pointers and sizes only; no exceptions cross the boundary; every entry point returns 0 or a
 * negative MI_DEGENSAC_E* code.  Host-pointer entry points stage through device memory themselves;
 * the *_dev entry points take device pointers (HBM-resident inputs/outputs) and a hipStream_t.
 * The whole estimation (sampling, minimal solver, scoring, DEGENSAC test, local optimisation,
 * adaptive termination, final mask) runs in one persistent HIP kernel; workgroups pull image pairs
 * from a device-wide ticket.
 * There is NO CPU fallback: without a usable gfx950 device every call fails with MI_DEGENSAC_ENODEV.
 *
 * Threading (SURVEY 8b "thread-safe handle/context"; the reference is not re-entrant: global
 * HASH_TABLE hash.h:32 + libc RNG).  Every entry point may be called from any number of host
 * threads at once:
 *   - a `mi_degensac_ctx` owns one HIP stream, pinned host staging and its device buffers; calls on
 *     ONE context are serialised by the caller, different contexts are independent;
 *   - the entry points without a context argument use a context private to the calling thread;
 *   - the *_dev entry points are asynchronous on the caller's stream; the per-launch scratch is
 *     private to (device, stream), so launches on different streams never share memory, and the
 *     thread's current HIP device is left as it was found;
 *   - enqueueing is serialised per DEVICE only (launches for different devices never wait for each other);
 *   - launches may overlap on one device, also with other processes' work: no workgroup of these kernels ever waits for
 *     a workgroup that may not have been dispatched.  (Cross-workgroup work — the cooperative mode for one very large
 *     pair, pairs that are set aside and resumed — is handed over through queues and claim counters: whoever is
 *     running takes the next unit, and a wait is only ever for a unit that a RUNNING workgroup has claimed.)
 */
#ifndef MI_DEGENSAC_H
#define MI_DEGENSAC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_DEGENSAC_OK        0
#define MI_DEGENSAC_EINVAL   -1   /* bad shape / argument (bindings.cpp:32-47, :267-282 -> ValueError) */
#define MI_DEGENSAC_ENODEV   -2   /* no HIP device / wrong architecture */
#define MI_DEGENSAC_EHIP     -3   /* HIP runtime error (see mi_degensac_last_error) */
#define MI_DEGENSAC_ENOMEM   -4

/* error_type values: utils.py:15-22, bindings.cpp:10-17 */
#define MI_DEGENSAC_F_SAMPSON        0
#define MI_DEGENSAC_F_SYMM_EPIPOLAR  1
#define MI_DEGENSAC_H_SAMPSON        0
#define MI_DEGENSAC_H_SYMM_SQ_MAX    1
#define MI_DEGENSAC_H_SYMM_MAX       2
#define MI_DEGENSAC_H_SYMM_SQ_SUM    3
#define MI_DEGENSAC_H_SYMM_SUM       4

/* flags */
#define MI_DEGENSAC_FLAG_FINAL_LAF_FILTER 1u  /* apply the F driver's final LAF filter, which the reference
                                                 guards with an uninitialised variable (exp_ranF.c:1254,1725) */

#define MI_DEGENSAC_FLAG_LEGACY_F 2u          /* fundamental matrix: behave like the reference's older drivers exp_ransacF (exp_ranF.c:242)
                                                 and exp_ransacFcustom (:811) instead of exp_ransacFcustomLAF: the sample budget
                                                 follows EVERY new best model, also one found by the main loop or the DEGENSAC
                                                 branch between two local optimisations (:1085-1090 sits outside the LO block
                                                 there).  No LAF check.  symmetric_error_check = 1 gives exp_ransacFcustom's own
                                                 symmetric check: over ALL points, against CHECK_COEF * th = 16 px_th^2, and in
                                                 the final mask against the model the driver computed LAST, not the best one
                                                 (exp_ranF.c:943-953, :1196-1203); error_type 0 without it = exp_ransacF */

/* per-call scheduling switches (results never depend on them).  They win over the process-wide defaults below
 * (mi_degensac_set_stream_mode / mi_degensac_set_hjob_mode and their environment variables): */
#define MI_DEGENSAC_FLAG_NO_STREAM   4u       /* fundamental matrix: no producer workgroups for this call (stream mode off)           */
#define MI_DEGENSAC_FLAG_STREAM_ON   8u       /* fundamental matrix: stream mode on whatever the batch size                            */
#define MI_DEGENSAC_FLAG_STREAM_TEST(b) (((uint32_t)(b) & 3u) << 8)   /* with STREAM_ON: bit 0 = the owner scores every chunk it takes from
                                                 the producer again, bit 1 = pairs ask for a producer even while unstarted pairs remain */
#define MI_DEGENSAC_FLAG_STREAM_AUTO 64u      /* fundamental matrix: the library's automatic choice for this call, whatever the process-wide default says \
 */
#define MI_DEGENSAC_FLAG_HJOB_ON     128u     /* homography: helper workgroups on for this call, whatever the process-wide default says \
 */
#define MI_DEGENSAC_FLAG_NO_HJOB     16u      /* homography: no helper workgroups for the local optimisations of this call             */

/* tuning word (0 = let the library decide; results never depend on it, only speed does):
 *   bits 0-1  kernel variant    1 = latency (512-thread workgroups)   2 = throughput (256-thread, 2 pairs per CU)
 *                               3 = high throughput (128-thread, up to 4 pairs per CU)
 *   bits 2-3  placement         1 = points + sampler pool in HBM      2 = both in LDS      3 = pool in LDS
 *             (a placement that does not fit the device's LDS is ignored)
 *   bit  4    sampler           1 = always use the sequential pool-swap stage
 *   bit  5    homography only: run the ten repetitions of every local optimisation one after the other on the whole
 *             workgroup instead of one repetition per wave (tests; the residual dump of
 *             mi_degensac_find_homography_resids always does).  EINVAL on a fundamental-matrix call
 *   bit  6    fundamental matrix with helper workgroups only (bits 8-15 or automatic): distribute EVERY pass over the whole
 *             point set over the claiming workgroups, not only those over >= 8192 points (tests).  EINVAL on a homography call
 *   bit  7    fundamental matrix only: run the repetitions of the local optimisation (exp_inFranicustom) and of innerH (the
 *             plane homography's local optimisation inside the DEGENSAC branch) one after the other on the whole workgroup
 *             instead of one repetition per wave from speculated generator states (tests).  EINVAL on a homography call
 *   bits 8-15 helper workgroups per pair of the cooperative large-n mode (fundamental matrix, placement HBM):
 *             0 = automatic (23 helpers when n >= 8192 and the batch leaves the device mostly idle, 15 with less room), 255 = off
 *   bits 16-23 setting pairs aside (fundamental matrix, batches larger than the resident grid): a pair still running
 *             after this many samples (units of 256) while unstarted pairs remain is written back to its workspace and
 *             queued; free workgroups take unstarted pairs first, then the queued pairs with many samples left, then
 *             the rest.  The first samples of every pair become a short discovery round after which the pairs with the
 *             most work left restart first, so the batch ends at about (sum of pair times) / (resident workgroups)
 *             instead of one long pair after the last pair was started.  0 = automatic (1024 samples), 255 = off
 *   bits 24-28 cap on the number of resident workgroups (0 = none, 1..31; for tests of the queueing paths on small batches)
 *   bits 29-31 with bits 16-23: a pair set aside with at least (threshold << this) samples left counts as "long" and is
 *             resumed before the others; 0 = automatic (threshold x 8).  EINVAL on a homography call
 * Every field has its own bits; a field that does not apply to the call (see above) is rejected with MI_DEGENSAC_EINVAL
 * rather than silently reinterpreted. */
#define MI_DEGENSAC_TUNE_VARIANT(v)   ((uint32_t)(v) & 3u)
#define MI_DEGENSAC_TUNE_PLACEMENT(p) (((uint32_t)(p) & 3u) << 2)
#define MI_DEGENSAC_TUNE_SEQ_POOL     (1u << 4)
#define MI_DEGENSAC_TUNE_H_SERIAL_LO  (1u << 5)
#define MI_DEGENSAC_TUNE_COOP_ALL_PASSES (1u << 6)
#define MI_DEGENSAC_TUNE_F_SERIAL_REPS (1u << 7)
#define MI_DEGENSAC_TUNE_HELPERS(h)   (((uint32_t)(h) & 255u) << 8)
#define MI_DEGENSAC_TUNE_SET_ASIDE(t)  (((uint32_t)(t) & 255u) << 16)
#define MI_DEGENSAC_TUNE_GRID_CAP(g)   (((uint32_t)(g) & 31u) << 24)
#define MI_DEGENSAC_TUNE_LONG_SHIFT(l) (((uint32_t)(l) & 7u) << 29)

typedef struct mi_degensac_params {
    double   px_th;                    /* pixel threshold (utils.py:76,113)                         */
    double   conf;                     /* confidence for adaptive termination                       */
    int32_t  max_iters;                /* hard cap on minimal samples                               */
    int32_t  error_type;               /* MI_DEGENSAC_F_* / MI_DEGENSAC_H_*                          */
    int32_t  symmetric_error_check;    /* bool                                                      */
    int32_t  enable_degeneracy_check;  /* bool, fundamental only (utils.py:119)                     */
    double   laf_consistensy_coef;     /* <=0: off; needs dim == 6                                  */
    uint32_t flags;
    uint32_t tuning;                   /* MI_DEGENSAC_TUNE_*; 0 = automatic                          */
} mi_degensac_params;

/* per-pair int32 statistics block (the reference computes most of these and drops them:
 * bindings.cpp:242-243, exp_ranF.c:1758-1759, exp_ranH.c:922-927) */
#define MI_DEGENSAC_STATS_LEN 16
enum {
    MI_ST_SAMPLES = 0,      /* minimal samples drawn (data_out[0])                                  */
    MI_ST_LO_RUNS = 1,      /* local optimisations run (data_out[1])                                */
    MI_ST_REJECTED = 2,     /* H: samples rejected before scoring (data_out[2]); F: candidates turned down by the LAF
                               consistency check (S.Ilafs < maxS.Ilafs, exp_ranF.c:1410, :1553, :1681)                 */
    MI_ST_I = 3,            /* inlier count of the returned model (driver return value)             */
    MI_ST_MODELS = 4,       /* models scored against all N points through the metric pointers       */
    MI_ST_DEGEN = 5,        /* F: DEGENSAC plane-and-parallax completions                           */
    MI_ST_IH = 6,           /* F: largest H-inlier count seen (exp_ranF.c *Ih)                      */
    MI_ST_BEST_SAMPLE = 7,  /* sample number at which the returned model was committed              */
    MI_ST_FULL_PASSES = 8, MI_ST_EX_PASSES = 9, MI_ST_H_PASSES = 10, MI_ST_AUX_PASSES = 11,
    MI_ST_TICKS_BEST = 12,  /* 100 MHz device wall-clock ticks from the pair's start to that commit; the time a pair waits
                               while it is set aside (batches) is not counted: ticks = time it was being worked on */
    MI_ST_TICKS_TOTAL = 13, /* ... to the pair's end                                                */
    MI_ST_THREADS = 14,     /* workgroup size of the kernel variant that ran (512 / 256 / 128)      */
    MI_ST_PLACEMENT = 15    /* bits 0-7: 0 = points + pool in HBM, 1 = both in LDS, 2 = pool in LDS; bit 8: the pair was set aside once;
                               bit 9: its chunks came from a producer workgroup (stream mode); bit 10: a hand-over wait of the launch
                               timed out before this pair ended and its results were DISCARDED (zero model, all-zero mask, I = 0) —
                               the asynchronous *_dev entry points report the failure this way, the host-pointer entry points run
                               such pairs again without producers / helpers and set bit 11 on the pairs they re-ran */
};

/* Stream mode (fundamental matrix): a pair hands the outcome-independent part of its main loop (sample stream, 7-point
 * solves, screening / scoring) to a workgroup that has run out of pairs; results never depend on it.  Every pair of a launch that
 * uses the mode (automatic: batches of at most two pairs per resident workgroup) posts its request with its first chunk; idle
 * workgroups take the request with the most samples left, the oldest first.  MI_DEGENSAC_STREAM_MIN_SAM=<n> (environment, read
 * once) makes pairs ask only after n samples; MI_DEGENSAC_STREAM_DEPTH=<chunks> bounds the ring.
 * mode: -1 = automatic (default), 0 = off, values > 0 = on with test bits in (mode >> 1): bit 0 = the owner scores every chunk
 * it takes from the producer again, bit 1 = pairs ask for a producer even while unstarted pairs remain.  This is the process-wide
 * DEFAULT (returns the previous one; environment: MI_DEGENSAC_STREAM); a call chooses for itself with MI_DEGENSAC_FLAG_NO_STREAM /
 * MI_DEGENSAC_FLAG_STREAM_ON in params.flags. */
int mi_degensac_set_stream_mode(int mode);      /* any mode > 0 turns the mode on; its test bits are (mode >> 1).  Default of the calls that
                                                   set none of MI_DEGENSAC_FLAG_NO_STREAM / _STREAM_ON / _STREAM_AUTO and run on a context
                                                   without its own setting (mi_degensac_ctx_set_scheduling): process-wide, kept for the
                                                   entry points that take no context */
/* Homography: workgroups that have run out of pairs take whole repetitions of the local optimisations of the pairs that still
 * run (results never depend on it).  1 = on (default), 0 = off.  Process-wide DEFAULT (returns the previous one; environment:
 * MI_DEGENSAC_HJOB); a call switches them off for itself with MI_DEGENSAC_FLAG_NO_HJOB in params.flags. */
int mi_degensac_set_hjob_mode(int mode);        /* default of the calls that do not set MI_DEGENSAC_FLAG_NO_HJOB */
/* Hand-overs between workgroups (stream mode, homography helpers, the cooperative large-n mode) never wait for a workgroup that
 * may not have been dispatched.  Every wait for a RUNNING workgroup — the stream mode's data waits, the owner's wait for the
 * repetitions its homography helpers claimed, the cooperative mode's waits for claimed units — gives up after the same limit, 4 s of
 * device wall clock (dg_wait_count / dg_stream_wait).  The launch then raises its error word, every pair that ends afterwards
 * discards its results (MI_ST_PLACEMENT bit 10: zero model, all-zero mask, I = 0), and the host-pointer entry points run those pairs
 * again in a second launch without producers / helpers (bit 11) instead of failing the call.  The asynchronous *_dev entry points
 * cannot do that: their ONLY report of a discarded pair is bit 10 of its stats block, so a caller that wants to tell "discarded" from
 * "no model found" must pass d_stats (with d_stats = NULL both read as a zero model).  (One wait is not a hand-over and has no limit:
 * an idle helper workgroup of the cooperative mode sleeping until its owner publishes the next stage or retires the slot.)
 * Test hook: the limit in 100 MHz ticks (0 = fault injection: every such wait fails at once; < 0 restores the default); returns the
 * previous limit. */
long long mi_degensac_set_wait_ticks(long long ticks);
/* Memory: the library caches one scratch buffer per (device, stream): one workspace per resident workgroup (about 0.65 MB at
 * n = 2000) plus, for fundamental-matrix batches larger than the resident grid, one spare workspace per queued pair (within half
 * of the free memory, at most 24 GB) and, in stream mode, one ring of at most 512 chunk entries of 13.5 KB per resident workgroup
 * (within a quarter of the free memory, at most 8 GB; about 2.7 GB at the default max_iters on a whole device).  A buffer more
 * than four times larger than eight launches in a row needed is given back; mi_degensac_release_scratch frees it at once. */

/* ---- contexts ---------------------------------------------------------------------------------- */
typedef struct mi_degensac_ctx mi_degensac_ctx;
int  mi_degensac_ctx_create(int device, mi_degensac_ctx **out);
void mi_degensac_ctx_destroy(mi_degensac_ctx *ctx);
/* the context's stream (hipStream_t), e.g. to order other work after a call */
void *mi_degensac_ctx_stream(mi_degensac_ctx *ctx);
/* Scheduling defaults of ONE context (results never depend on them): every call on `ctx` whose params.flags say nothing about the stream
 * mode / the homography helpers takes these instead of the process-wide defaults of mi_degensac_set_stream_mode / _set_hjob_mode (which
 * remain for the entry points without a context).  stream_mode: -2 = inherit the process-wide default (initial), -1 = automatic,
 * 0 = off, > 0 = on with test bits in (mode >> 1).  hjob_mode: -2 = inherit (initial), 0 = off, 1 = on.  Returns 0 or MI_DEGENSAC_EINVAL. */
int mi_degensac_ctx_set_scheduling(mi_degensac_ctx *ctx, int stream_mode, int hjob_mode);

/* ---- host-pointer entry points (mirror the pybind signatures) -------------------------------- */
/* pts1, pts2: [n, dim] row-major float64, dim in {2, 6}; F/H: 9 doubles row-major as the reference's C
 * driver returns them (H is the internal image2->image1 column-wise form; the Python layer applies
 * inv(H.T), utils.py:108); mask: n bytes (0/1); stats: MI_DEGENSAC_STATS_LEN int32 or NULL. */
int mi_degensac_find_fundamental(const double *pts1, const double *pts2, int n, int dim,
                                 const mi_degensac_params *prm, uint32_t seed, int device,
                                 double *F, uint8_t *mask, int32_t *stats);
int mi_degensac_find_homography(const double *pts1, const double *pts2, int n, int dim,
                                const mi_degensac_params *prm, uint32_t seed, int device,
                                double *H, uint8_t *mask, int32_t *stats);

/* ---- batches of independent pairs (ragged): pair p owns rows offsets[p] .. offsets[p+1] ------- */
int mi_degensac_find_fundamental_batch(const double *pts1, const double *pts2, const int64_t *offsets,
                                       int n_pairs, int dim, const mi_degensac_params *prm,
                                       const uint32_t *seeds, int device,
                                       double *F /*[n_pairs*9]*/, uint8_t *mask /*[offsets[n_pairs]]*/,
                                       int32_t *stats /*[n_pairs*16] or NULL*/);
int mi_degensac_find_homography_batch(const double *pts1, const double *pts2, const int64_t *offsets,
                                      int n_pairs, int dim, const mi_degensac_params *prm,
                                      const uint32_t *seeds, int device,
                                      double *H, uint8_t *mask, int32_t *stats);
/* ONE batch over several devices from one process (no collective: the host gathers): device devices[k] gets the k-th contiguous
 * block of pairs (blocks as even as possible, the first n_pairs % n_devices hold one pair more), one host thread per block;
 * seeds travel with their pairs, so results do not depend on the device list.  A device may be listed more than once. */
int mi_degensac_find_fundamental_batch_multi(const double *pts1, const double *pts2, const int64_t *offsets, int n_pairs, int dim,
                                             const mi_degensac_params *prm, const uint32_t *seeds, const int *devices, int n_devices,
                                             double *F, uint8_t *mask, int32_t *stats);
int mi_degensac_find_homography_batch_multi(const double *pts1, const double *pts2, const int64_t *offsets, int n_pairs, int dim,
                                            const mi_degensac_params *prm, const uint32_t *seeds, const int *devices, int n_devices,
                                            double *H, uint8_t *mask, int32_t *stats);
/* the same on an explicit context (its device, its stream, its staging buffers); blocking */
int mi_degensac_ctx_find_fundamental_batch(mi_degensac_ctx *ctx, const double *pts1, const double *pts2,
                                           const int64_t *offsets, int n_pairs, int dim,
                                           const mi_degensac_params *prm, const uint32_t *seeds,
                                           double *F, uint8_t *mask, int32_t *stats);
int mi_degensac_ctx_find_homography_batch(mi_degensac_ctx *ctx, const double *pts1, const double *pts2,
                                          const int64_t *offsets, int n_pairs, int dim,
                                          const mi_degensac_params *prm, const uint32_t *seeds,
                                          double *H, uint8_t *mask, int32_t *stats);

/* ---- device-pointer entry points: everything except `offsets_host` and `prm` lives in HBM ----- */
/* `stream` is a hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing; never
 * synchronises the device. */
int mi_degensac_find_fundamental_batch_dev(const double *d_pts1, const double *d_pts2,
                                           const int64_t *d_offsets, const int64_t *offsets_host,
                                           int n_pairs, int dim, const mi_degensac_params *prm,
                                           const uint32_t *d_seeds, int device, void *stream,
                                           double *d_F, uint8_t *d_mask, int32_t *d_stats);
int mi_degensac_find_homography_batch_dev(const double *d_pts1, const double *d_pts2,
                                          const int64_t *d_offsets, const int64_t *offsets_host,
                                          int n_pairs, int dim, const mi_degensac_params *prm,
                                          const uint32_t *d_seeds, int device, void *stream,
                                          double *d_H, uint8_t *d_mask, int32_t *d_stats);
/* ---- RANSAC on ellipse-to-ellipse correspondences: the reference's ransacH2el (degensac/ranH2el.h:35, ranH2el.c:19-206;
 * no binding in the reference's Python layer — this is its C signature with the allocation-free device conventions of
 * the entry points above).  u10: [total, 10] doubles, per correspondence x1 y1 a1 b1 c1 | x2 y2 a2 b2 c2 with the local
 * affine frame [a 0; b c] of each image; a sample is TWO correspondences (14 x 15 system of Chum & Matas, ICPR 2012).
 * H (9 doubles per pair, column-wise like the reference's internal H) maps image 2 to image 1; mask[j] = HDs residual of
 * the returned model <= th (ranH2el.c:189-198).  stats as for the other drivers ([0] samples, [1] LO runs, [3] I). */
typedef struct mi_degensac_h2el_params {
    double   th;          /* threshold on the squared transfer error HDs returns (the reference's `th`)       */
    double   conf;        /* confidence of the sample-count rule (nsamples with sample size 2)                  */
    int32_t  max_iters;   /* max_sam                                                                            */
    int32_t  do_lo;       /* bool: local optimisation (inHraniEl)                                               */
    int32_t  inl_limit;   /* inlLimit: correspondences per least-squares fit inside the LO; 0 = all (>= 4 else)  */
    int32_t  reserved;
} mi_degensac_h2el_params;
int mi_degensac_ransac_h2el_batch(const double *u10, const int64_t *offsets, int n_pairs,
                                  const mi_degensac_h2el_params *prm, const uint32_t *seeds, int device,
                                  double *H, uint8_t *mask, int32_t *stats);
int mi_degensac_ransac_h2el_batch_dev(const double *d_u10, const int64_t *d_offsets, const int64_t *offsets_host, int n_pairs,
                                      const mi_degensac_h2el_params *prm, const uint32_t *d_seeds, int device, void *stream,
                                      double *d_H, uint8_t *d_mask, int32_t *d_stats);

/* release the scratch cached for (device, stream) pairs whose work has completed (all of them when
 * `stream_or_null` is NULL, else only that stream's); call before destroying a stream */
int mi_degensac_release_scratch(int device, void *stream_or_null);

/* ---- structured diagnostics ----------------------------------------------------------------
 * The reference's drivers fill a residual dump per local-optimisation run and the binding throws it away
 * (`double *resids`, exp_ranF.c:1503-1511 / :776-779 / :670-672 / :729-730, exp_ranH.c:679-698 / :445-446 / :344 /
 * :400; freed at bindings.cpp:242-243, :458-459).  Per LO run RESIDS_M = 62 rows of n residuals (rtools.h:15):
 *   row 0      the so-far-the-best sample's model (errs[4])        row 1      the least-squares model before the LO
 *   row 2+6i   repetition i: the model of its random subset        rows 3+6i .. 6+6i  its inner iterations (LO metric)
 *   row 7+6i   its final full-metric pass
 * resids[(pair_offset * resid_runs + run * n) * 62 + row * n + j]; rows the reference never writes for a run (early
 * exits: it leaves them uninitialised) are NaN here, rows it memsets keep its byte pattern (0 for F, 0xFF for H); the
 * after-loop LO of the F driver leaves row 0 NaN.  LO runs beyond resid_runs are not recorded. */
#define MI_DEGENSAC_RESIDS_M 62
typedef struct mi_degensac_diag {
    double  *d_resids;        /* device buffer of total_points * resid_runs * 62 doubles, or NULL              */
    int32_t  resid_runs;      /* LO runs per pair the buffer has room for                                     */
    int32_t  struct_size;     /* sizeof(mi_degensac_diag) of the caller's header.  0 (what callers built against the first layout
                                 pass in this field, then called `reserved`) = the layout that ends at d_hist: d_screen is NOT read.
                                 Fields added later are only read when struct_size covers them.                                */
    int32_t *d_hist;          /* fundamental matrix only: the drivers' `data_out` (exp_ranF.c:1495, :1758-1759; allocated and freed at
                                 bindings.cpp:412, :459): per pair n + 3 ints at d_hist[offsets[pair] + 3 * pair]: [0] samples drawn,
                                 [1] LO runs, [2 + I] number of samples whose best model had I inliers.  Device buffer of
                                 total_points + 3 * n_pairs ints, or NULL.  Every model is scored exactly when this is set (the
                                 screening passes are skipped): results are unchanged, the call is slower.               */
    int32_t *d_screen;        /* fundamental matrix only: per pair 4 ints at d_screen[4 * pair]: the models of the main loop's scoring phase
                                 (every chunk the pair's own workgroup scored, including samples speculated past the final budget) by the
                                 arithmetic they got: [0] entered the level-1 screen (single precision, two points per instruction),
                                 [1] entered the level-2 screen (double precision, the point's own denominator), [2] scored with the exact
                                 metric in the reference's operation order, [3] all of them.  MI_ST_MODELS counts every one of [3] (minus
                                 the speculated tail) as a reference-equivalent "model scored" although only [2] ran the exact arithmetic
                                 over all points.  Device buffer of 4 * n_pairs ints, or NULL; costs nothing measurable.            */
} mi_degensac_diag;
int mi_degensac_find_fundamental_batch_dev_ex(const double *d_pts1, const double *d_pts2, const int64_t *d_offsets,
        const int64_t *offsets_host, int n_pairs, int dim, const mi_degensac_params *prm, const uint32_t *d_seeds, int device,
        void *stream, double *d_F, uint8_t *d_mask, int32_t *d_stats, const mi_degensac_diag *diag /*nullable*/);
int mi_degensac_find_homography_batch_dev_ex(const double *d_pts1, const double *d_pts2, const int64_t *d_offsets,
        const int64_t *offsets_host, int n_pairs, int dim, const mi_degensac_params *prm, const uint32_t *d_seeds, int device,
        void *stream, double *d_H, uint8_t *d_mask, int32_t *d_stats, const mi_degensac_diag *diag /*nullable*/);
/* one pair, host pointers: resids[resid_runs * 62 * n] */
int mi_degensac_find_fundamental_resids(const double *pts1, const double *pts2, int n, int dim, const mi_degensac_params *prm,
        uint32_t seed, int device, double *F, uint8_t *mask, int32_t *stats /*nullable*/, double *resids, int resid_runs);
int mi_degensac_find_homography_resids(const double *pts1, const double *pts2, int n, int dim, const mi_degensac_params *prm,
        uint32_t seed, int device, double *H, uint8_t *mask, int32_t *stats /*nullable*/, double *resids, int resid_runs);
/* one pair, host pointers: hist[n + 3] (see mi_degensac_diag.d_hist) */
int mi_degensac_find_fundamental_hist(const double *pts1, const double *pts2, int n, int dim, const mi_degensac_params *prm,
        uint32_t seed, int device, double *F, uint8_t *mask, int32_t *stats /*nullable*/, int32_t *hist);

/* ---- tentative correspondences: the stage in front of the estimators -------------------
 * Replaces the matcher calls of the reference's example, examples/simple-example.py:46-53
 * (cv2.BFMatcher().knnMatch(descs1, descs2, k=2) followed by `m.distance < 0.9 * n.distance`):
 * brute-force 2-nearest-neighbour search of every row of desc1 among the rows of desc2, the
 * second-nearest-neighbour ratio test and an optional mutual-nearest-neighbour check.
 *   norm L2:      float32 descriptors [n, dim]; distance = sqrt(sum_k (a_k - b_k)^2), fp32, summed in ascending k
 *   norm HAMMING: uint8 descriptors [n, dim] with dim % 4 == 0 (pad with zero bytes); distance = differing bits
 * idx[i] = the two nearest train rows of query i (ties: lower index first; -1 when desc2 has fewer rows),
 * dist[i] their distances, keep[i] = dist[i][0] < ratio * dist[i][1] (and, with mutual, the nearest neighbour of
 * desc2[idx[i][0]] in desc1 is i). */
#define MI_DEGENSAC_NORM_L2       0
#define MI_DEGENSAC_NORM_HAMMING  1
int mi_degensac_match(int norm, const void *desc1, int n1, const void *desc2, int n2, int dim, float ratio, int mutual,
                      int device, int32_t *idx /*[n1,2]*/, float *dist /*[n1,2]*/, uint8_t *keep /*[n1], nullable*/);
/* device pointers, asynchronous on `stream` */
int mi_degensac_match_knn2_dev(int norm, const void *d_desc1, int n1, const void *d_desc2, int n2, int dim, int device,
                               void *stream, int32_t *d_idx, float *d_dist);
int mi_degensac_match_filter_dev(const int32_t *d_idx, const float *d_dist, int n1, float ratio,
                                 const int32_t *d_back_idx_or_null /*[n2,2]: knn2 of desc2 in desc1*/, int device, void *stream,
                                 uint8_t *d_keep);
/* utils.py:24-41 convert_cv2_kpts_to_xyA on the device: kpts [n,4] float32 = (pt.x, pt.y, size, angle in degrees) ->
 * out [n,6] float64 = (x, y, s cos a, s sin a, -s sin a, s cos a), the LAF rows of the estimators' [n,6] input */
int mi_degensac_kpts_to_xyA(const float *kpts, int n, int device, double *out);
int mi_degensac_kpts_to_xyA_dev(const float *d_kpts, int n, int device, void *stream, double *d_out);
const char *mi_degensac_match_last_error(void);

/* ---- unit-level device entry points (parity tests of the kernels' building blocks) ------------ */
/* score n_models fundamental (kind 0: Sampson, 1: symmetric epipolar) or homography (kind 10..14:
 * H Sampson, symm_sq_max, symm_max, symm_sq_sum, symm_sum) models against all n points: I (<= th)
 * and MSAC J per model, optionally the residual vectors [n_models*n]. Host pointers. */
int mi_degensac_score_models(const double *pts1, const double *pts2, int n, int dim,
                             const double *models, int n_models, int kind, double th, int device,
                             uint32_t *I, double *J, double *resid /*nullable*/);
/* the main-loop sample stream: for `iters` iterations the 7 (or 4) drawn ids in draw order.
 * seq_pool != 0 forces the sequential pool-swap stage. */
int mi_degensac_sample_stream(uint32_t seed, int n, int sample_size, int iters, int device,
                              int32_t *samples /*[iters*sample_size]*/);
int mi_degensac_sample_stream_ex(uint32_t seed, int n, int sample_size, int iters, int device,
                                 int seq_pool, int32_t *samples);
/* the 7-point solver + oriented-epipolar test on given samples: for each of n_samples 7-tuples of
 * ids (draw order) nsol[i] in 0..3 valid models (-1: null space not 2-dimensional) and up to 3 models
 * (27 doubles per sample) with the index of the cubic's root each came from. */
int mi_degensac_solve7(const double *pts1, const double *pts2, int n, int dim,
                       const int32_t *samples, int n_samples, int device,
                       int32_t *nsol, int32_t *root_idx /*[3*n_samples]*/, double *models /*[27*n_samples]*/);
/* the lane-level 3x3 routines of the DEGENSAC branch, one problem per lane: op 0 = in-place inverse (matutls/minv.c
 * order; in/out 9 doubles, flag = -1 when singular), op 1 = right singular vectors and singular values (matutls/svduv.c
 * order; in 9, out 9 + 3), op 2 = Hdetect (DegUtils.c:84-161; in F (9) + seven correspondences x1 y1 x2 y2 (28) + the
 * triplet as three doubles, out 9).  op 3 = the 9x9 symmetric eigen-solver (LAPACK dsyev as lap_eig calls it,
 * degensac/lapwrap.c:67-96), one problem per wave: in 81, out 9 eigenvalues (smallest first, the rest as the QL/QR
 * iteration left them) + 81 (column-major vectors, column 0 = the one of the smallest eigenvalue), flag = info.
 * op 4 = the real roots of a cubic as the 7-point solver takes them (Ftools.c:251-298; in 4 coefficients, out 3, flag =
 * the number of roots): the one routine of the path that calls the math library (pow / acos / cos).  The device takes their
 * correctly rounded values (dg_crmath.h), which the host's libm returns in all but ~0.1-0.2 % of its calls: the one place
 * where the device can differ from a host run of the reference in the last bits.
 * op 5 = op 3 with TWO problems per wave (dg_eig2.h: problem 2t in lanes 0..31 of wave t, 2t + 1 in lanes 32..63), same output.
 * op 6 / 7 = timing of op 3 / op 5: every wave solves its problem(s) flag[0] times (on entry) and leaves the 100 MHz ticks that took
 * in out[wave]. */
int mi_degensac_mat3(int op, const double *in, int count, int device, double *out, int32_t *flag);
/* the screening counts of the scoring phase (dg_score_tiles.h) for given fundamental-matrix models over a point set:
 * c1[m] = level-1 count (single precision, loosest denominator, rounding-widened threshold), c2[m] = level-2 count
 * (double precision, the point's own denominator, threshold x (1 + 1e-6)); both must be >= the number of points whose
 * exact residual is < 9/4 th (kind 0 = Sampson, 1 = symmetric epipolar).  Models in batches of 64 per wave, as in the kernels. */
int mi_degensac_screen_counts(const double *pts1, const double *pts2, int n, int dim, const double *models, int n_models,
                              int kind, double th, int device, uint32_t *c1, uint32_t *c2);

/* the homography main loop's screen (dg_geom.h dg_HDs_maybe_below, swept by dg_h_screen4 four models per wave): cnt[m] = points
 * that may lie below 9/4 th under the Sampson metric of model m (division-free bound, threshold x (1 + 1e-6)); cand (nullable,
 * [n_models * n]) = the per-point verdicts.  Every point whose exact HDs residual (Htools.c:161-200) is < 9/4 th must be a
 * candidate, and cnt[m] must equal the number of candidates. */
int mi_degensac_screen_counts_h(const double *pts1, const double *pts2, int n, int dim, const double *models, int n_models,
                                double th, int device, uint32_t *cnt, uint8_t *cand);

/* the wave forms of the reference's srand() / rand() (one multiply-add per state word instead of 341 dependent steps; up to 31
 * outputs per step as three interleaved prefix sums): out[i] = the (skip + i + 1)-th rand() after srand(seed), generated `block`
 * (1..31) values at a time.  Must equal libc's sequence. */
int mi_degensac_rng_wave(uint32_t seed, int skip, int block, int count, int device, int32_t *out);

/* Where the time of ONE host-pointer call goes (the single-call latency of bench.py's `single_call_ms.breakdown`).  With timing on,
 * every host-pointer call of the calling thread leaves, in milliseconds: [0] the whole call (host clock), [1] packing the inputs into
 * the pinned block, [2] enqueueing (H2D copy, scratch set-up, launch, D2H copy), [3] waiting for the stream, [4] unpacking the
 * results; and from HIP events on the context's stream: [5] the H2D copy, [6] header memsets + kernel + error-word copy, [7] the
 * D2H copy.  Process-wide switch (returns the previous value), per-thread result; off by default (it costs four events per call). */
#define MI_DEGENSAC_TIMING_LEN 8
int mi_degensac_set_call_timing(int on);
int mi_degensac_last_call_timing(double *out /*[MI_DEGENSAC_TIMING_LEN]*/);

/* ---- misc -------------------------------------------------------------------------------------- */
int         mi_degensac_device_count(void);
const char *mi_degensac_last_error(void);
const char *mi_degensac_version(void);
/* name of the dominant kernel (for profiling) */
const char *mi_degensac_kernel_name(int homography);
/* result of the one-time LDS exchange-order self-check of `device` (1 = the parallel pool-swap stage is in use,
 * 0 = the check failed and the sequential stage is used, <0 = error) */
int         mi_degensac_pool_stage_parallel(int device);

#ifdef __cplusplus
}
#endif
#endif /* MI_DEGENSAC_H */
