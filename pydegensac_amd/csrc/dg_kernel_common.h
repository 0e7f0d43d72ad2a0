/* Shared definitions of the persistent kernels: kernel arguments, per-pair workspace layout,
 * the device RNG fast path, the sampler and the lane-0 small LSQ solvers. */
#ifndef DG_KERNEL_COMMON_H
#define DG_KERNEL_COMMON_H
#include "dg_geom.h"

#ifndef DG_CHUNK
#define DG_CHUNK   256         /* minimal samples speculated per round (one per lane of the first DG_CHUNK); a variant may choose less */
#endif
/* phase timers of the development build (tools/gpu_phases.py): a real-time-counter read plus a wait on the scalar
 * memory queue each, on the critical path of the serial waves: compiled out of the product library */
#ifdef MI_DEGENSAC_DEV
#define DG_CLK() wall_clock64()
#define DG_DEVT(x) x
#else
#define DG_CLK() 0ll
#define DG_DEVT(x) do {} while (0)
#endif
#ifndef DG_MINW
#define DG_MINW    2           /* __launch_bounds__: minimum waves per SIMD the kernels are compiled for (register budget 512 / DG_MINW) */
#endif
#define DG_MCAP    96          /* models scored per LDS sub-batch                                  */
#define DG_HT_CAP  4096        /* LO inlier-set hash entries per pair                              */
/* one table of chunk models, [3 * DG_CHUNK][9] doubles; the cooperative mode has two (+ DG_PRE_BYTES: what the early solves of the
 * next chunk leave per sample) */
#define DG_MTAB_BYTES ((((size_t)3 * DG_CHUNK * 9 * sizeof(double)) + 255) & ~(size_t)255)
#define DG_PRE_BYTES  ((size_t)DG_CHUNK * 2 * sizeof(int))

/* rtools.h:4-15 */
#define DG_ITER_SAM 50
#define DG_RAN_REP 10
#define DG_ILSQ_ITERS 4
#define DG_TC 4
#define DG_MWM (9/4)

struct dg_params {
    double th;            /* squared / linear threshold as the C driver receives it               */
    double sym_th;        /* SymCheck_th                                                          */
    double laf_coef;
    double conf;
    int    max_iters;
    int    error_type;
    int    degen;
    int    final_laf_filter;
    int    h2_do_lo, h2_inl_limit;   /* ransacH2el (dg_kernel_h2el.h): do_lo and inlLimit of ranH2el.h:35 (0 = no limit) */
    int    legacy;        /* F: the sample-budget rule of exp_ransacF / exp_ransacFcustom (MI_DEGENSAC_FLAG_LEGACY_F) */
};

/* per-pair global scratch (L2-resident), sized for the pair's n */
struct dg_ws_layout {
    size_t stride;        /* bytes per pair                                                       */
    size_t off_lists;     /* 6 int lists of n_max                                                 */
    size_t off_flags;     /* 4 byte-flag vectors of n_max                                         */
    size_t off_ht;        /* hash table: heads[64] + entries[DG_HT_CAP][4] ints                    */
    size_t off_models;    /* chunk models: [3*DG_CHUNK][9] doubles + tags                          */
    size_t off_pts;       /* dg_pt[n_max] when the points do not fit LDS                          */
    size_t off_pool;      /* int[n_max]   ditto                                                   */
    size_t off_stage;     /* dg_pt[2 * n_max]: staging of long least-squares lists + their per-coordinate arrays */
    size_t off_wave;      /* per-wave buffers of the wave-parallel sections: int[NW][n_max] + dg_pt[NW][n_max] */
    size_t off_res;       /* per-model results of a chunk: double J[768], unsigned I[768], int rf[256][5] */
    size_t off_mslot;     /* cooperative mode: the chunk's compact model index -> slot table, unsigned short[3*DG_CHUNK]; then the per-model
                             screening counters unsigned[3*DG_CHUNK] and the survivor list unsigned short[3*DG_CHUNK] */
    size_t off_park;      /* parked-pair image: dg_f_shared + the dynamic LDS of the workgroup that set the pair aside (F driver) */
    size_t off_hjbuf;     /* cooperative mode: ordered MSAC terms of each claiming workgroup, double[coop_k + 1][n_max]   */
    size_t off_job;       /* cooperative mode: the job of a distributed pass (dg_coop_job), its per-slice records and the slice-local
                             staging of its outputs: int[n_max] x 2 (lists), double[n_max] (MSAC terms)                       */
    size_t off_hrep;      /* homography LO, one repetition per wave: dg_hrep_log[DG_RAN_REP], then per wave int[2][n_max] (id lists),
                             8 n_max bytes of slack (the MSAC terms go through LDS; keeps what follows 16-byte aligned for any n_max),
                             dg_pt[2 * n_max] (the gathered points of its long least-squares lists; the second half is spare) */
    int    hrep;          /* 1 = that area exists */
    int    n_max;
};

/* What one repetition of exp_inHranicustom (exp_ranH.c:415-467) leaves behind when it is run on its own by one wave: every
 * model whose residuals the reference would have put into an errs[] buffer, and every (hash, I, J) that decides something.
 * The repetitions only depend on each other through the inlier-set hash table, the best-so-far score and the rotation of
 * the errs[] buffers; dg_inHranic_waves replays those in repetition order from these records. */
struct dg_hrep_it { double hl[9]; double J; unsigned hash; int I; };
/* (one record per 128-byte-aligned block: repetitions of one job may be written by workgroups on different XCDs, whose L2s are
 * not coherent with each other; two records in one cache line could lose one writer's bytes at write-back) */
struct alignas(128) dg_hrep_log {
    double h0[9]; double J0; int I0;       /* the sample's model and its score at th */
    int nit;                               /* iterations of exp_iterHcustom that scored their model (0..4) */
    int last_short;                        /* 1: the last of them left fewer than 4 ids for the next fit (exp_ranH.c:366) */
    int has_fin;                           /* 1: the model after the last iteration was scored (exp_ranH.c:397-408) */
    dg_hrep_it it[4];
    double hf[9]; double Jf; int If;
    int ids[12];                           /* the sample */
};

/* Cooperative large-n mode (placement HBM, fundamental matrix): every pair owner has coop_k helper workgroups that
 * share the scoring of the current chunk's models against the point set in the owner's HBM workspace.  One control block
 * per owner slot (zeroed by the host per launch).  The owner publishes a STAGE (plain parameter stores, one agent-scope
 * release, then a new `gen`): stage 1 = tile-major screening counts (unit = a slice of the point set, every wave of the
 * claiming workgroup runs its share of the models over it and adds its counts to the per-model device counters), stage 2
 * = exact scoring of the survivors (unit = one model, scored by the whole claiming workgroup).  Units are CLAIMED from a
 * generation-tagged counter with compare-and-swap, by the helpers and — once its sampler stages are done — by the owner
 * itself, so a helper that is not resident (another launch holds its CU) simply never claims anything: nobody waits for
 * a workgroup that has not been dispatched, and the grid does not have to be co-resident.  `tau_bits` is the pair's
 * best-score bound (min of the two running best MSAC gains, as ordered bits of a non-negative double): only the owner
 * raises it (atomic max; it never falls while a pair runs), every claiming workgroup reads it per unit, and a model whose
 * screening count does not exceed it is dropped without exact scoring.  (MI355X_MICROARCH.md, inter-workgroup visibility:
 * plain payload, lane-0 agent release / acquire, relaxed agent-scope flags on their own 128-byte line.) */
struct dg_coop_cb {
    /* line 0: flags, touched ONLY with agent-scope atomics (a plain access would leave a copy of the line in the
     * toucher's XCD L2, which later polls would hit: per-XCD L2s are not coherent with each other) */
    int gen;              /* stage generation, -1 = no more work for this slot's helpers */
    int done;             /* units of generation `gen` that are finished */
    int next;             /* (gen & 0xfffff) << 12 | next unclaimed unit of that generation */
    int fpad0;
    unsigned long long tau_bits;   /* device-wide monotone best-score bound of the running pair */
    int fpad[26];
    /* line 1: the stage's parameters, plain stores before the owner's release, plain loads after a successful claim (the
     * stage cannot end before every claimed unit is done, so they are stable while a claimer reads them) */
    int stage, n_units, Mtot, n, kind, slice, use_l1, err;
    double th, ext[4];
    int mtab, mpad;       /* which of the two model tables holds the chunk that is being scored (the other one receives the next chunk's solves) */
    double ppad[6];
};
static_assert(sizeof(dg_coop_cb) == 256, "control block = two 128-byte lines");

/* Stage 3 of the cooperative mode: ONE pass of the local optimisation over the whole point set (residuals of model F under
 * metric `kind`, inlier count, MSAC terms, up to two ordered inlier lists), distributed over point slices.  A unit = one
 * slice: the claiming workgroup runs the ordinary workgroup pass on it and leaves the slice's lists and nonzero MSAC terms
 * in slice-local staging (at the slice's own offset) and its counts in rec[unit]; the owner concatenates the lists in slice
 * order (= point order) and adds the terms one after the other in that order: the same lists and the same J as one
 * workgroup would produce. */
struct dg_coop_job {
    double F[9], thJ, thL, thL2;
    int kind, wantJ, listStrict, has_list, has_list2, slice, n, pad;
};
struct dg_coop_rec { unsigned I, nL, nL2, nJ; };
#define DG_COOP_MAX_SLICES 64

/* Stream mode (fundamental matrix, placements LDS / pool-in-LDS): a pair with many samples left gets a PRODUCER workgroup — any
 * workgroup of the launch that has run out of pairs — which takes over the outcome-independent part of the pair's main loop:
 * the sample stream (seed chain, draws, pool swaps), the 7-point solves and the screening / scoring of every chunk, with a
 * bound tau it reads from the owner.  The owner keeps the pair's state and only commits the chunks in order (and runs the
 * local optimisations and the DEGENSAC branch they trigger), so its events overlap the sampling instead of alternating with
 * it.  Hand-over = the image the owner writes for setting a pair aside (the complete state between two chunks, wl.off_park):
 * the producer resumes that image in producer mode, skips the chunks the owner has meanwhile committed (sampler stages only)
 * and then leaves one dg_stream_ent per chunk in the owner's ring.  One control block per resident workgroup (owner slot).
 * Flags on the first 128-byte line, touched with agent-scope atomics only; the payload (image, ring entries) is plain memory
 * behind the usual release / acquire pair (MI355X_MICROARCH.md, inter-workgroup visibility). */
enum { DG_ST_IDLE = 0, DG_ST_REQ = 1, DG_ST_BUSY = 2, DG_ST_ATTACHED = 3, DG_ST_RELEASED = 4 };
struct dg_stream_cb {
    /* DG_ST_*: IDLE -> REQ (image valid) -> ATTACHED (a producer took it) -> RELEASED (producer gone) -> IDLE; BUSY = the owner rewrites the image */
    int state;
    int head;                 /* chunks the producer has published: sequence numbers < head are in the ring */
    int tail;                 /* chunks the owner has consumed */
    int stop;                 /* the owner is done with the pair */
    /* since when the pair has been worked on (wall_clock64 >> 10): producers take the oldest of the requests with the most samples left */
    int owner_sam;
    int max_sam;              /* the owner's current sample budget */
    unsigned long long tau_bits;   /* the owner's current bound min(maxS.J, maxSs.J), as the bits of a double */
    int claim;                /* fan mode: next chunk (sequence number) a worker may claim; head = chunks whose drawn ids are in the ring */
    int fpad[23];
    /* second line: written by the owner before the release that opens the request */
    int pair, wsid, img_sam, fan_kind;        /* fan mode: the metric of the main loop's scoring (DG_K_*) */
    double fan_th, fan_ext[4];                /* fan mode: threshold and coordinate extents of the pair (the screens' bounds) */
    int ppad[18];
};
static_assert(sizeof(dg_stream_cb) == 256, "stream control block = two 128-byte lines");

/* Homography kernel: the ten repetitions of a local optimisation (exp_ranH.c:415-467) are independent jobs once their samples are
 * drawn (dg_inHranic_waves): every wave claims whole repetitions from this block — the owner's waves, and the waves of HELPER
 * workgroups, i.e. workgroups of the launch that have run out of pairs.  A helper reads the pair's points, the sample and the
 * hash table in the owner's workspace (read-only while the job is open), works on its own staging area and LDS, and leaves
 * the repetition's 600-byte record (dg_hrep_log) in the owner's workspace; the owner replays the records in order.
 * Flags on the first 128-byte line (agent-scope atomics only), the job's parameters on the second. */
struct dg_hjob_cb {
    int gen;                  /* generation of the open job, 0 = none */
    int next;                 /* (gen << 8) | repetitions claimed so far */
    int done;                 /* repetitions finished */
    int fpad[29];
    int n, kind, ssiz, wsid; double th; int ppad[26];
};
static_assert(sizeof(dg_hjob_cb) == 256, "job control block = two 128-byte lines");

struct dg_args {
    const double *pts1, *pts2;       /* [total, dim] */
    const long long *offsets;        /* [n_pairs + 1] */
    const unsigned *seeds;           /* [n_pairs] */
    double *model_out;               /* [n_pairs, 9] */
    unsigned char *mask_out;         /* [total] */
    int *stats_out;                  /* [n_pairs, 16] or null */
    char *ws;                        /* per-SLOT scratch (slot = resident workgroup), wl.stride bytes each */
    dg_ws_layout wl;
    dg_params prm;
    int dim, n_pairs, pts_in_lds;
    int dev_knob;                    /* development: one integer for A/B experiments (mi_degensac_dev_set_knob); 0 = defaults.  Sits in padding */
    double *resids_out;              /* optional diagnostics: the reference's per-LO residual dump (exp_ranF.c:1503-1511, :776-779,
                                        :670-672, :729-730; exp_ranH.c likewise), [offsets[pair] * resid_runs * 62 + (run * 62 + row) * n + j];
                                        rows the reference never writes for a run are NaN; null = off                        */
    int resid_runs;                  /* LO runs per pair the dump has room for                                   */
    int lo_width;                    /* fundamental matrix: repetitions of a local optimisation started per round of speculated generator states
                                        (0 = one per wave; sits in the padding behind resid_runs) */
    int *hist_out;                   /* optional diagnostics (F driver): the reference's data_out (exp_ranF.c:1495, :1758-1759): per pair
                                        n + 3 ints at hist_out[offsets[pair] + 3 * pair]: [0] samples, [1] LO runs, [2 + I] = number of samples
                                        whose best root had I inliers.  Needs every model scored exactly: switches the screens off. null = off */
    int *screen_out;                 /* optional diagnostics (F driver): [n_pairs, 4] main-loop models by arithmetic (dg_f_shared::scnt); null = off */
    int *ticket;                     /* device counter, zeroed per launch: persistent workgroups pull the next pair from it */
    const int *order;                /* optional processing order (ticket t -> pair order[t]); null = identity              */
    int coop_k;                      /* helper workgroups per owner (0 = every workgroup owns pairs)            */
    int coop_pass_min;               /* cooperative mode: passes over at least this many points are distributed (stage 3) */
    dg_coop_cb *coop;                /* [owner slots] control blocks (coop_k > 0)                              */
    /* Setting pairs aside (F driver, batches larger than the resident grid).  Which pairs run long is only known while
     * they run, and a batch ends one long pair after the last long pair was started.  So every pair that is still running
     * after park_sam samples while unstarted pairs remain is written back to its workspace (the LDS image goes to
     * wl.off_park) and queued by what it has left: max_sam - no_sam >= park_long -> queue 1, else queue 0; its workgroup
     * continues with a spare workspace.  A free workgroup takes, in this order: the next unstarted pair, queue 1, queue 0.
     * The first park_sam samples of every pair are thus a cheap discovery round, after which the pairs with the most work
     * left are (re)started first (longest-processing-time-first on discovered information); the batch ends at about
     * (sum of pair times) / (resident workgroups) instead of one long pair after the last start.
     * Results do not depend on it (the image is the complete state between two chunks). */
    int park_sam;                    /* 0 = off */
    int n_res, n_ws;                 /* workspaces in `ws`: n_res = one per resident workgroup, then the spares up to n_ws */
    int dyn_bytes;                   /* dynamic LDS per workgroup                                              */
    int *park_ctl;                   /* [0] spares handed out; queue q (0 = few samples left, 1 = many): [32 + 64 q] entries claimed,
                                        [64 + 64 q] entries taken (one 128-byte line each) */
    long long *park_q;               /* [2][park_cap] entries: pair << 32 | workspace, -1 until published        */
    int park_cap;                    /* entries per queue                                                       */
    int park_long;                   /* a pair set aside with at least this many samples left goes to queue 1    */
    int lo_serial;                   /* homography: 1 = run the repetitions of a local optimisation one after the other on the whole workgroup */
    int pool_seq;                    /* 1 = always use the sequential pool-swap stage (LDS exchange-order self-check failed, or forced) */
    /* fundamental matrix: 1 = the repetitions of innerH and of the local optimisation one after the other on the whole workgroup (tests) */
    int innerh_serial;
    int variant_threads, mode;       /* reported in the stats block */
    /* stream mode (dg_stream_cb): */
    int stream_on;                   /* 0 = off */
    int stream_early;                /* 1: the launch has a spare workgroup per pair from the start (small batches, single calls): pairs ask at once */
    int stream_min_sam, stream_min_left;   /* a pair asks for a producer once it has drawn stream_min_sam samples and has at least stream_min_left left */
    int stream_depth;                /* ring entries per owner */
    int stream_test;                 /* bit 0: the owner re-scores every chunk it takes from the ring (tests the stale-bound path); bit 1: ask for a producer
                                        whether or not unstarted pairs remain */
    size_t stream_ent_bytes;         /* bytes per ring entry (>= sizeof(dg_stream_ent) of every variant) */
    dg_stream_cb *scb;               /* [n_res] */
    char *ring;                      /* [n_res][stream_depth] entries */
    int *done_pairs;                 /* [0] pairs finished (header): workgroups without work leave when it reaches n_pairs; [1] open producer requests
                                        (stream mode); [2] open local-optimisation jobs (homography helpers) */
    dg_hjob_cb *hjob;                /* homography: [n_res] job control blocks, or null (no helper workgroups) */
    /* Fan mode (fundamental matrix, one large pair per owner, everything in the HBM workspace; dg_f_fan.h): the owner draws the sample stream
     * and commits; fan_k WORKER workgroups per owner claim whole chunks of drawn ids from the owner's ring, solve and score them against the
     * owner's points and leave the stream mode's chunk entries; the cooperative helpers keep the local optimisations' stages. */
    int fan_k;                       /* worker workgroups per owner; 0 = off */
    int fan_ws0;                     /* workspace index of the first worker (worker w of owner o: fan_ws0 + o * fan_k + w) */
    int *fan_flags;                  /* [n_res][stream_depth] "entry done" words, one 128-byte line each (agent-scope atomics only): seq + 1 */
    int *err_flag;                   /* set when a hand-over wait times out: every pair that ends afterwards discards its results (zero model, zero mask,
                                        bit 10 of stats[15]) and the host-pointer entry points run those pairs again without helpers (dg_discard_if_failed) */
    int *trace;                      /* debug: [0] = count, then (tag, I, J lo, J hi) records; null = off */
    int trace_cap;
    int wait_ticks;                  /* limit of the stream mode's data waits in 100 MHz ticks (4 s; the test hook mi_degensac_set_wait_ticks shortens it).
                                        Sits in the padding behind trace_cap: the homography kernel at 256 threads has exactly 80 KB of static LDS
                                        (this block is copied into LDS), and one more 8-byte field halves its residency */
    long long *phase_out;            /* debug: [n_pairs][8] 100 MHz ticks per phase (sample, solve, score, commit+events, LO, degen, tail, total) */
};

/* ONE wave, uniform control flow: wait until the agent-scope counter *cnt reaches `target`, for at most A.wait_ticks of the 100 MHz clock
 * (4 s; mi_degensac_set_wait_ticks).  Every wait of an owner for units / repetitions that RUNNING workgroups have claimed goes through
 * here (cooperative large-n mode, homography helpers): a wait that long is a bug or a stuck device — raise the launch's error word
 * (`code`), give up and go on; every pair that ends afterwards discards its results and the host-pointer entry points run it again
 * without helpers.  wait_ticks == 0 is the test hook's fault injection: the wait fails at once.  Returns false on a time-out. */
__device__ __forceinline__ bool dg_wait_count(const dg_args &A, int *cnt, int target, int code, int sleep_ticks)
{
    const long long t0 = wall_clock64(), limit = (long long)A.wait_ticks;
    for (;;) {
        const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= target && limit != 0) return true;
        if (wall_clock64() - t0 > limit || limit == 0) {
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(A.err_flag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (sleep_ticks >= 8) __builtin_amdgcn_s_sleep(8); else if (sleep_ticks >= 4) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(2);
    }
}

/* ---- glibc TYPE_3 fast path ----------------------------------------------------------------------
 * After srandom(seed) the k-th output is ((sum_j C[k][j] * r_j) mod 2^32) >> 1 with r_0 = seed,
 * r_j = 16807 * r_{j-1} mod (2^31-1): the 310 discarded steps of r[i] = r[i-31] + r[i-3] are linear
 * over Z/2^32, so they collapse into the constant 8 x 31 matrix C (filled by the host at load time
 * by running the generator on unit vectors).  G[j] = 16807^j mod (2^31-1). */
static __constant__ unsigned dg_rng_C[8][32];
static __constant__ unsigned dg_rng_Ct[32][8];     /* transposed copy: one 32-byte scalar load per term j */
static __constant__ unsigned dg_rng_G[32];
/* [j][p]: coefficient (mod 2^32) of the initial word r_j in ring word p after srandom's 310 discarded steps */
static __constant__ unsigned dg_rng_T[31][32];

__device__ __forceinline__ unsigned dg_mulmod31(unsigned a, unsigned b)
{
    unsigned long long x = (unsigned long long)a * b;
    x = (x & 0x7fffffffull) + (x >> 31);
    x = (x & 0x7fffffffull) + (x >> 31);
    if (x >= 0x7fffffffull) x -= 0x7fffffffull;
    return (unsigned)x;
}
/* first LCG step exactly as glibc does it (handles seeds >= 2^31, i.e. negative int32) */
__device__ __forceinline__ unsigned dg_lcg_first(int r0)
{
    /* for 0 <= r0 < 2^31 Schrage's step equals 16807*r0 mod (2^31-1) exactly */
    if (r0 >= 0) return dg_mulmod31((unsigned)r0, 16807u);
    long long hi = r0 / 127773, lo = r0 % 127773;
    long long word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    return (unsigned)word;
}
/* all 8 outputs after srand(seed): o[0..6] raw draws, o[7] the next seed */
__device__ __forceinline__ void dg_rng_outputs(unsigned seed, unsigned *o)
{
    if (seed == 0) seed = 1;
    unsigned r1 = dg_lcg_first((int)seed);
    unsigned acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = dg_rng_Ct[0][k] * seed;
    (void)r1;
    for (int j = 1; j < 31; j++) {
        unsigned rj = dg_mulmod31(seed < 0x7fffffffu ? seed : r1, seed < 0x7fffffffu ? dg_rng_G[j] : dg_rng_G[j - 1]);
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += dg_rng_Ct[j][k] * rj;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = acc[k] >> 1;
}
/* wave-cooperative: next seed only (lanes 0..30 carry one term each) */
__device__ __forceinline__ unsigned dg_rng_next_seed_wave(unsigned seed, int lane, int kout = 7)
{
    if (seed == 0) seed = 1;
    unsigned r1 = dg_lcg_first((int)seed);
    unsigned rj = (lane == 0) ? seed : ((lane < 31) ? dg_mulmod31(r1, dg_rng_G[(lane - 1) & 31]) : 0u);
    unsigned term = (lane < 31) ? dg_rng_C[kout][lane & 31] * rj : 0u;
    return dg_wave_sum_u(term) >> 1;
}

/* ---- the generator on one wave ---------------------------------------------------------------------------------
 * dg_srand / dg_rand (dg_dev_small.h) are the reference's libc calls step by step on one lane: 341 dependent steps per srand, a
 * few dependent LDS round trips per rand — 50 us and 0.15 us.  Both are linear over Z/2^32, so a wave does them at once:
 *   dg_srand_wave: ring word p after the 310 discarded steps = sum_j T[j][p] r_j with r_0 = seed, r_j = r_1 16807^(j-1) mod (2^31-1)
 *                  (T filled by the host from unit vectors, like dg_rng_C): one multiply-add per term, lane p owns word p;
 *   dg_rand_block: the next n <= 31 outputs.  With s[0..30] the ring in chronological order (s[0] = r[f] is 31 steps old, s[28] = r[b]),
 *                  o_1 = s[0] + s[28], o_2 = s[1] + s[29], o_3 = s[2] + s[30], o_t = s[t-1] + o_(t-3): three interleaved prefix sums,
 *                  i.e. o_(j+1) = (s[j] + s[j-3] + s[j-6] + ...) + s[28 + j mod 3]; lane j < n stores it where step j + 1 would.
 * Same state, same outputs as the step-by-step calls (tests/test_gpu_units.py::test_wave_generator_equals_libc). */
__device__ __forceinline__ void dg_srand_wave(dg_rng *g, unsigned seed, int lane)
{
    if (seed == 0) seed = 1;
    const unsigned r1 = dg_lcg_first((int)seed);
    const unsigned rj = lane == 0 ? seed : (lane < 31 ? dg_mulmod31(r1, dg_rng_G[(lane - 1) & 31]) : 0u);
    const int p = lane < 31 ? lane : 0;
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 31; j++) acc += dg_rng_T[j][p] * (unsigned)__builtin_amdgcn_readlane((int)rj, j);
    DG_WSYNC();
    if (lane < 31) g->r[lane] = (int32_t)acc;
    if (lane == 0) { g->f = 3; g->b = 0; }             /* 310 = 10 x 31 steps: the ring pointers are back where srandom put them */
    DG_WSYNC();
}
/* all 64 lanes of one wave, n <= 31: lane i < n returns what the (i + 1)-th dg_rand(g) from here would; the state moves n steps */
__device__ __forceinline__ int dg_rand_block(dg_rng *g, int n, int lane)
{
    const int f = g->f, b = g->b;
    int idx = f + (lane < 31 ? lane : 30); if (idx >= 31) idx -= 31;
    const unsigned s = (unsigned)g->r[idx];
    unsigned x = lane < 31 ? s : 0u, t;
    t = (unsigned)__shfl_up((int)x, 3, 64);  if (lane >= 3)  x += t;
    t = (unsigned)__shfl_up((int)x, 6, 64);  if (lane >= 6)  x += t;
    t = (unsigned)__shfl_up((int)x, 12, 64); if (lane >= 12) x += t;
    t = (unsigned)__shfl_up((int)x, 24, 64); if (lane >= 24) x += t;
    const unsigned o = x + (unsigned)__shfl((int)s, 28 + lane % 3, 64);
    DG_WSYNC();                                        /* every lane has read the ring before any lane writes it */
    if (lane < n) g->r[idx] = (int32_t)o;
    if (lane == 0) { int nf = f + n, nb = b + n; g->f = nf >= 31 ? nf - 31 : nf; g->b = nb >= 31 ? nb - 31 : nb; }
    DG_WSYNC();
    return (int)(o >> 1);
}
/* all 64 lanes of one wave: n calls of dg_rand(g) whose values nobody reads */
__device__ __forceinline__ void dg_rand_skip(dg_rng *g, int n, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n);
    while (n > 0) { const int m = n > 31 ? 31 : n; (void)dg_rand_block(g, m, lane); n -= m; }
}

/* ---- LO hash table: the reference's 64 chained buckets (hash.c:49-96, hash.h:21-32) ------------ */
struct dg_ht { int *heads; int *ent; int *count; };     /* ent: [cap][4] = hash, length, iterID, next */
__device__ __forceinline__ void dg_ht_init(dg_ht &h, int tid)
{
    if (tid < 64) h.heads[tid] = -1;
    if (tid == 0) *h.count = 0;
}
__device__ __forceinline__ int dg_ht_contains(const dg_ht &h, unsigned hash, int length, int iterID)
{
    int e = h.heads[hash % 64];
    while (e >= 0) { if ((unsigned)h.ent[4*e] == hash && h.ent[4*e+1] == length && h.ent[4*e+2] == iterID) return iterID; e = h.ent[4*e+3]; }
    e = h.heads[hash % 64];
    while (e >= 0) { if ((unsigned)h.ent[4*e] == hash && h.ent[4*e+1] == length) return h.ent[4*e+2]; e = h.ent[4*e+3]; }
    return -1;
}
/* all 64 lanes of one wave: is the set (hash, length) in the table under ANY iterID?  The same answer as dg_ht_contains(.., -1) != -1 —
 * every entry with this hash sits in the chain of bucket hash % 64, so looking at all entries finds exactly what the two chain walks find —
 * from count / 64 independent loads per lane instead of a walk of dependent ones by one lane (an L2 round trip per step). */
__device__ __forceinline__ bool dg_ht_known_wave(const dg_ht &h, unsigned hash, int length, int lane)
{
    typedef __attribute__((address_space(1))) int dg_gint;
    const dg_gint *ent = (const dg_gint *)h.ent;
    const int cnt = __builtin_amdgcn_readfirstlane(*(const dg_gint *)h.count);
    bool hit = false;
    for (int e = lane; e < cnt; e += 64) hit = hit || ((unsigned)ent[4*e] == hash && ent[4*e+1] == length);
    return __ballot(hit) != 0ull;
}
__device__ __forceinline__ void dg_ht_insert(dg_ht &h, unsigned hash, int length, int iterID)
{
    int e = *h.count;
    if (e >= DG_HT_CAP) return;
    h.ent[4*e] = (int)hash; h.ent[4*e+1] = length; h.ent[4*e+2] = iterID; h.ent[4*e+3] = h.heads[hash % 64];
    h.heads[hash % 64] = e; *h.count = e + 1;
}

/* ---- lane-0 helpers on global int lists --------------------------------------------------------- */
/* rtools.c:25-39 randsubset on a list in global memory; returns the offset of the subset (max_sz - siz) */
__device__ __forceinline__ int dg_randsubset(dg_rng *g, int *pool, int max_sz, int siz)
{
    for (int i = 0; i < siz; i++) {
        int s = dg_rand(g) % (max_sz - i);
        int j = max_sz - i - 1;
        int q = pool[s]; pool[s] = pool[j]; pool[j] = q;
    }
    return max_sz - siz;
}


/* Wave-cooperative randsubset (all 64 lanes of one wave call it; siz <= 32).  The draws depend only on the RNG,
 * so lane 0 makes them first; the <= 2*siz list positions involved (tail slots T_i = max_sz-1-i in lanes 0..siz-1,
 * drawn slots s_i in lanes siz..2*siz-1) are then loaded with ONE parallel access, the swaps are replayed on
 * registers (first lane holding a position is its canonical slot), and the canonical slots are stored back with
 * one parallel access: two memory round trips instead of 4*siz dependent ones.  Same list contents afterwards.
 * *id = for lane j < siz, the id at subset position j (list[max_sz - siz + j]).  Returns the subset's offset. */
__device__ __forceinline__ int dg_randsubset_wave(dg_rng *g, int *pool, int max_sz, int siz, int lane, int *id)
{
    /* the siz draws at once (dg_rand_block, <= 31 of them): lane i holds draw i */
    int myS = 0;
    if (siz <= 31) { const int raw_ = dg_rand_block(g, siz, lane); if (lane < siz) myS = raw_ % (max_sz - lane); }
    else for (int i = 0; i < siz; i++) {
        int s = 0;
        if (lane == 0) s = dg_rand(g) % (max_sz - i);
        s = __builtin_amdgcn_readfirstlane(s);
        if (lane == i) myS = s;
    }
    const bool used = lane < 2 * siz;
    const int drawn = __shfl(myS, lane >= siz ? lane - siz : 0, 64);
    const int pos = lane < siz ? max_sz - 1 - lane : (used ? drawn : -1);
    int val = used ? pool[pos] : 0;
    for (int i = 0; i < siz; i++) {
        const int s_i = __builtin_amdgcn_readlane(myS, i), t_i = max_sz - 1 - i;
        const unsigned long long mA = __ballot(used && pos == s_i), mB = __ballot(used && pos == t_i);
        const int la = __ffsll((long long)mA) - 1, lb = __ffsll((long long)mB) - 1;
        const int va = __builtin_amdgcn_readlane(val, la), vb = __builtin_amdgcn_readlane(val, lb);
        if (lane == la) val = vb;
        if (lane == lb) val = va;
    }
    bool canon = used;
    for (int j = 0; j < 2 * siz; j++) { const int pj = __builtin_amdgcn_readlane(pos, j); if (j < lane && pj == pos) canon = false; }
    if (canon) pool[pos] = val;
    /* subset position j <-> tail slot T_{siz-1-j}; tail lanes are always canonical */
    *id = __shfl(val, lane < siz ? siz - 1 - lane : 0, 64);
    return max_sz - siz;
}

/* dg_randsubset_wave on a private generator, leaving the list untouched: the canonical slots it would store are
 * written to (pos[j], val[j]), j < 2*siz, pos = -1 for the slots that hold no store.  Used to prepare a sample ahead of
 * time; storing the slots later (and adopting the generator) has the same effect as the call itself. */
__device__ __forceinline__ void dg_randsubset_wave_ahead(dg_rng *g, const int *pool, int max_sz, int siz, int lane, int *id, int *pos_out, int *val_out)
{
    /* the siz draws at once (dg_rand_block, <= 31 of them): lane i holds draw i */
    int myS = 0;
    if (siz <= 31) { const int raw_ = dg_rand_block(g, siz, lane); if (lane < siz) myS = raw_ % (max_sz - lane); }
    else for (int i = 0; i < siz; i++) {
        int s = 0;
        if (lane == 0) s = dg_rand(g) % (max_sz - i);
        s = __builtin_amdgcn_readfirstlane(s);
        if (lane == i) myS = s;
    }
    const bool used = lane < 2 * siz;
    const int drawn = __shfl(myS, lane >= siz ? lane - siz : 0, 64);
    const int pos = lane < siz ? max_sz - 1 - lane : (used ? drawn : -1);
    int val = used ? pool[pos] : 0;
    for (int i = 0; i < siz; i++) {
        const int s_i = __builtin_amdgcn_readlane(myS, i), t_i = max_sz - 1 - i;
        const unsigned long long mA = __ballot(used && pos == s_i), mB = __ballot(used && pos == t_i);
        const int la = __ffsll((long long)mA) - 1, lb = __ffsll((long long)mB) - 1;
        const int va = __builtin_amdgcn_readlane(val, la), vb = __builtin_amdgcn_readlane(val, lb);
        if (lane == la) val = vb;
        if (lane == lb) val = va;
    }
    bool canon = used;
    for (int j = 0; j < 2 * siz; j++) { const int pj = __builtin_amdgcn_readlane(pos, j); if (j < lane && pj == pos) canon = false; }
    if (used) { pos_out[lane] = canon ? pos : -1; val_out[lane] = val; }
    *id = __shfl(val, lane < siz ? siz - 1 - lane : 0, 64);
}

#endif /* DG_KERNEL_COMMON_H */
