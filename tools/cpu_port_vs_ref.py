"""CPU check: oracle port vs the unmodified reference build (oracle/_ref) on randomised seeded problems — F and H, every
metric, LAF on/off, degenerate scenes, n = 8 .. 3000: sample / LO counters, mask and model must be identical.
    python tools/cpu_port_vs_ref.py [n_cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import port, ref
from pydegensac_amd import synthetic as syn


LAF_STATS = [0, 0]          # F cases with the LAF check on; of those, cases in which it turned a candidate down


def run(N, rng_seed, verbose=True):
    """Returns (cases whose mask or counters differ, cases whose model differs by > 1e-9, worst model difference)."""
    rng = np.random.default_rng(rng_seed)
    bad = 0; skipped = 0; loose = 0; worst = 0.0
    for case in range(N):
        seed = int(rng.integers(1, 2**31 - 1)); n = int(rng.choice([8, 20, 64, 150, 400, 1000, 2000, 3000])); mi = int(rng.choice([500, 3000, 20000]))
        if rng.random() < 0.6:
            ir = float(rng.uniform(0.1, 0.8)); sg = float(rng.choice([0.05, 0.1, 0.5, 1.0])); pf = float(rng.choice([0.0, 0.0, 0.6, 0.9]))
            et = int(rng.choice([0, 1])); sym = bool(rng.random() < 0.7); dg = bool(rng.random() < 0.7); th = float(rng.choice([0.5, 1.0, 2.0]))
            laf = bool(rng.random() < 0.4); lc = float(rng.choice([1.0, 2.0, 3.0])) if laf else 0.0; lbad = float(rng.choice([0.1, 0.25, 0.5]))
            p1, p2, _, _ = syn.two_view_fundamental(n, ir, sg, seed=case, plane_fraction=pf, laf=laf, laf_bad=lbad, laf_sigma=float(rng.choice([0.05, 0.5])))
            Mp, mp, sp = port.find_fundamental(p1, p2, th, 0.9999, mi, et, sym, lc, dg, seed=seed)
            Mr, mr, sr = ref.find_fundamental(p1, p2, th, 0.9999, mi, et, sym, lc, dg, seed=seed)
            tag = f"F n={n} ir={ir:.2f} sig={sg} pf={pf} et={et} sym={sym} dg={dg} th={th} mi={mi} laf={lc} bad={lbad} laf_rej={sp['rejected']}"
            LAF_STATS[0] += laf; LAF_STATS[1] += sp["rejected"] > 0
        else:
            if n < 12:
                n = 12                          # n <= 10 goes through the reference's 4-point u2h path (uninitialised reads)
            ir = float(rng.uniform(0.15, 0.8)); sg = float(rng.choice([0.2, 0.5, 1.0])); laf = bool(rng.random() < 0.5)
            et = int(rng.integers(0, 5)); sym = bool(rng.random() < 0.7); th = float(rng.choice([1.0, 2.0, 4.0])); lc = 3.0 if laf else 0.0
            p1, p2, _, _ = syn.homography_pairs(n, ir, sg, seed=case, laf=laf)
            Mp, mp, sp = port.find_homography(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed)
            Mr, mr, sr = ref.find_homography(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed)
            tag = f"H n={n} ir={ir:.2f} sig={sg} laf={laf} et={et} sym={sym} th={th} mi={mi}"
        Mp = np.asarray(Mp, float).ravel(); Mr = np.asarray(Mr, float).ravel()
        if not np.isfinite(Mr).all() or np.abs(Mr).sum() == 0 or sr["I"] == 0:
            # no model found: the reference returns uninitialised memory for model and mask; compare counters only
            ok = sp["samples"] == sr["samples"] and sp["lo_runs"] == sr["lo_runs"]; skipped += 1
        else:
            rel = np.linalg.norm(Mp - Mr) / max(np.linalg.norm(Mr), 1e-300)
            ok = np.array_equal(np.asarray(mp, bool), np.asarray(mr, bool)) and sp["samples"] == sr["samples"] and sp["lo_runs"] == sr["lo_runs"]
            if ok and rel >= 1e-9:
                # same trajectory and mask, model off by more than rounding: an ill-conditioned final LSQ (plane-dominated
                # scene), where LAPACK's dsyev (OpenBLAS here) and the restated netlib dsyev split a near-multiple eigenvalue
                loose += 1; worst = max(worst, rel)
                if rel >= 1e-6 and verbose: print("model differs", f"{rel:.2e}", tag, "seed", seed)
        if not ok:
            bad += 1
            if verbose: print("MISMATCH", tag, "seed", seed, "port", sp["samples"], sp["lo_runs"], sp["I"], "ref", sr["samples"], sr["lo_runs"], sr["I"])
    return bad, loose, worst, skipped


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.time()
    bad, loose, worst, skipped = run(N, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{N - bad}/{N} identical masks and counters ({skipped} without a model: counters only); model beyond 1e-9 in {loose} "
          f"(worst {worst:.2e}) in {time.time() - t0:.0f} s; F + LAF cases {LAF_STATS[0]}, with LAF rejections {LAF_STATS[1]}")
