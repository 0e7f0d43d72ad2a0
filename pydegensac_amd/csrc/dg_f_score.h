/* Scoring phase of the main loop, one wave: the chunk's models through the tile-major screens of dg_score_tiles.h, the survivors through the exact
 * metric with the reference's sequential MSAC sum.
 * Part of the fundamental-matrix kernel: included by dg_kernel_f_main.h, in this order, after dg_kernel_f.h and dg_score_tiles.h. */
#ifndef DG_F_SCORE_H
#define DG_F_SCORE_H
#define DG_AS3L(T) __attribute__((address_space(3))) T

/* ---------------------------------------------------------------------------------------------- */
/* Scoring phase of one wave (own register allocation).  The chunk's models are dealt round-robin to the NS scoring
 * waves (wave ws takes mi = ws, ws + NS, ...); lane j of the wave owns the wave's j-th model of the current batch of up
 * to 64.  Screens (dg_score_tiles.h): level 1 when tau >= 64, level 2 when tau >= 4, each tile-major over the whole
 * point set with the models' coefficients in this wave's LDS table `tab`; models whose count does not exceed tau get
 * J = 0 (never an event in the commit, so decisions are unchanged); the survivors are scored exactly, one wave per
 * model: I, and J as the reference's sequential sum (dg_seq_sum): the wave stores the nonzero terms in point order, lane 0
 * adds them one after the other. */
template <int LDSPTS>
__device__ __noinline__ void dg_score_chunk_F(const dg_pt *P, int n, const double *gmodels, const unsigned short *mslot,
                                             int Mtot, int ws, int NS, int kind, double th, double tauJ, const double *ext /* LDS[4] */,
                                             char *tab /* LDS, this wave's */, int tab_bytes,
                                             double *jbuf /* this wave's scratch, >= n doubles */, unsigned *res_I, double *res_J, int lane,
                                             unsigned *scnt /* LDS[4]: dg_f_shared::scnt; null = do not count (a second scoring of the same chunk) */,
                                             int m_first = 0 /* only the models mi >= m_first (the commit's re-scoring of the rest of a chunk) */,
                                             bool allow_l1 = true /* false: level 2 only (large point sets with few inliers: random models have far more
                                                                     points inside the looser level-1 band than the bound to beat) */)
{
    /* workgroup-uniform arguments arrive in vector registers (separate function): make the loop control scalar again */
    n = __builtin_amdgcn_readfirstlane(n); Mtot = __builtin_amdgcn_readfirstlane(Mtot); ws = __builtin_amdgcn_readfirstlane(ws);
    NS = __builtin_amdgcn_readfirstlane(NS); kind = __builtin_amdgcn_readfirstlane(kind); tab_bytes = __builtin_amdgcn_readfirstlane(tab_bytes);
    m_first = __builtin_amdgcn_readfirstlane(m_first);
    const double t94 = th * 9 / 4, t94b = t94 * (1.0 + 1e-6);
    const bool use_bound = th != 0 && kind != DG_K_EXFSYM && tauJ >= 4.0;
    const bool use_l1 = use_bound && allow_l1 && tauJ >= 64.0;
    const int nm = Mtot > ws ? (Mtot - ws + NS - 1) / NS : 0;
    int B1 = tab_bytes / (DG_L1_ENTRY_FLOATS * (int)sizeof(float)), B2 = tab_bytes / (DG_L2_ENTRY_DOUBLES * (int)sizeof(double));
    B1 = B1 > 64 ? 64 : B1; B2 = B2 > 64 ? 64 : B2;
    float *tab_f = (float *)tab; double *tab_d = (double *)tab;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < nm; j0 += 64) {
        const int nb = nm - j0 < 64 ? nm - j0 : 64;
        const int mi = ws + (j0 + lane) * NS;                       /* this lane's model (have) */
        const bool have = lane < nb && mi >= m_first;
        if (__ballot(have) == 0ull) continue;
        double F[9];
        {
            const double *gp = gmodels + (size_t)mslot[have ? mi : ws] * 9;
#pragma unroll
            for (int j = 0; j < 9; j++) F[j] = gp[j];
        }
        unsigned long long surv = __ballot(have);
        unsigned n_all = (unsigned)nb, n_l1 = 0, n_l2 = 0;
        if (use_l1) {
            n_l1 = (unsigned)nb;
            unsigned C1 = 0;
            for (int s0 = 0; s0 < nb; s0 += B1) {
                const int sb = nb - s0 < B1 ? nb - s0 : B1;
                const bool in = lane >= s0 && lane < s0 + sb;
                if (in) {
                    float Ff[9]; const float thr = dg_l1_setup(kind, F, ext, t94b, Ff);
                    float *e = tab_f + (lane - s0) * DG_L1_ENTRY_FLOATS;
#pragma unroll
                    for (int j = 0; j < 9; j++) e[j] = Ff[j];
                    e[9] = thr; e[10] = 0.f; e[11] = 0.f;
                }
                DG_WSYNC();
                const unsigned cq = dg_l1_tile_counts<LDSPTS>(P, 0, n, tab_f, sb, lane);      /* lane r < sb: model s0 + r */
                const unsigned cs = (unsigned)__shfl((int)cq, (lane - s0) & 63, 64);
                if (in) C1 = cs;
                DG_WSYNC();
            }
            const bool keep = have && ((double)C1 > tauJ);
            if (have && !keep) { res_I[mi] = 0; res_J[mi] = 0; }
            surv = __ballot(keep);
        }
        if (use_bound && surv) {
            const bool mine = (surv >> lane) & 1ull;
            const int myrank = __popcll(surv & lt_mask), ns = __popcll(surv);
            n_l2 = (unsigned)ns;
            unsigned C2 = 0;
            for (int s0 = 0; s0 < ns; s0 += B2) {
                const int sb = ns - s0 < B2 ? ns - s0 : B2;
                const bool in = mine && myrank >= s0 && myrank < s0 + sb;
                if (in) {
                    double *e = tab_d + (myrank - s0) * DG_L2_ENTRY_DOUBLES;
#pragma unroll
                    for (int j = 0; j < 9; j++) e[j] = F[j];
                    e[9] = 0.;
                }
                DG_WSYNC();
                const unsigned cq = dg_l2_tile_counts<LDSPTS>(P, 0, n, tab_d, sb, kind, t94b, lane);   /* lane r < sb: survivor s0 + r */
                const unsigned cs = (unsigned)__shfl((int)cq, (myrank - s0) & 63, 64);
                if (in) C2 = cs;
                DG_WSYNC();
            }
            const bool keep = mine && ((double)C2 > tauJ);
            if (mine && !keep) { res_I[mi] = 0; res_J[mi] = 0; }
            surv = __ballot(keep);
        }
        if (lane == 0 && scnt) {         /* four LDS adds per batch of up to 64 models */
            __hip_atomic_fetch_add(&scnt[0], n_l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&scnt[1], n_l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&scnt[2], (unsigned)__popcll(surv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&scnt[3], n_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        for (unsigned long long m = surv; m; m &= m - 1ull) {
            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            const int mie = ws + (j0 + l) * NS;
            double Fe[9];
#pragma unroll
            for (int j = 0; j < 9; j++) Fe[j] = dg_readlane_d(F[j], l);
            /* J: the nonzero terms of a step go, in point order, through this wave's LDS table (the batch's screens are done with it) and are
             * added before the next step, as in the local optimisation's passes (dg_wpass_impl) — they used to be written to the wave's
             * global buffer and read back by lane 0 sixteen at a time, one L2 round trip per sixteen dependent adds.  A table too small for
             * a whole step (the six scoring waves of the 512-thread kernel share the block) is flushed after every G tiles. */
            unsigned cI = 0, sJ = 0;
            double J = 0.0;
            DG_AS3L(double) *tl = (DG_AS3L(double) *)(double *)tab;
            const int G = __builtin_amdgcn_readfirstlane(tab_bytes / (64 * (int)sizeof(double)) < DG_PU ? tab_bytes / (64 * (int)sizeof(double)) : DG_PU);
            for (int base = 0; base < n; base += 64 * DG_PU) {
                dg_pt qq[DG_PU]; double dd[DG_PU];
#pragma unroll
                for (int u = 0; u < DG_PU; u++) { const int p = base + 64 * u + lane; qq[u] = dg_ldpt<LDSPTS>(P, p < n ? p : 0); }
#pragma unroll
                for (int u = 0; u < DG_PU; u++) dd[u] = dg_Ferr(kind, Fe, qq[u]);
#pragma unroll
                for (int u = 0; u < DG_PU; u++) {
                    const bool act = base + 64 * u + lane < n; const double d = dd[u];
                    double term = 0.0; if (act && th != 0 && !(d >= t94)) term = 1 - (d / t94);
                    cI += (act && d <= th) ? 1u : 0u;
                    const bool nz = !(term == 0.0);
                    const unsigned long long bJ = __ballot(nz);
                    if (G > 0) { if (nz) tl[sJ + (unsigned)__popcll(bJ & lt_mask)] = term; }
                    else if (nz) ((__attribute__((address_space(1))) double *)jbuf)[sJ + (unsigned)__popcll(bJ & lt_mask)] = term;
                    sJ += (unsigned)__popcll(bJ);
                    if (G > 0 && ((u + 1) % G == 0 || u == DG_PU - 1)) {
                        DG_WSYNC_LDS();
                        if (sJ) J = dg_seq_sum_impl<3>((const double *)tab, (int)sJ, J);
                        sJ = 0;
                        DG_WSYNC_LDS();
                    }
                }
            }
            DG_WSYNC();
            if (G == 0) { if (lane == 0) J = dg_seq_sum(jbuf, (int)sJ); J = __shfl(J, 0, 64); }
            const unsigned I = dg_wave_sum_u(cI);
            DG_WSYNC();
            if (lane == 0) { res_I[mie] = I; res_J[mie] = J; }
        }
    }
}

#endif /* DG_F_SCORE_H */
