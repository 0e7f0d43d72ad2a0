"""GPU: helper workgroups of the homography kernel (dg_hjob_cb).  The ten repetitions of a local optimisation are
jobs claimed by waves: the owner's, and those of workgroups that have run out of pairs, which read the pair's points, sample
and hash table in the owner's workspace and leave the repetition's record there.  Results must not depend on who ran a
repetition: with helpers on and off, every workgroup size, against the CPU oracle pair by pair."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu

VARIANT = {512: _lib.TUNE_LATENCY, 256: _lib.TUNE_THROUGHPUT, 128: _lib.TUNE_THROUGHPUT4}


@pytest.fixture()
def hjob_mode():
    prev = _lib.set_hjob_mode(1)
    yield
    _lib.set_hjob_mode(prev)


def _check(ora, H, m, st, tag):
    for p, (Ho, mo, so) in enumerate(ora):
        assert (st[p]["samples"], st[p]["lo_runs"], st[p]["rejected"], st[p]["models"], st[p]["best_sample"]) == \
               (so["samples"], so["lo_runs"], so["rejected"], so["models"], so["best_sample"]), (tag, p, st[p], so)
        assert np.array_equal(np.asarray(m[p]), mo), (tag, p)
        if np.abs(Ho).sum() > 0:
            Hu = np.linalg.inv(np.asarray(Ho).T)
            assert np.linalg.norm(np.asarray(H[p]) - Hu) <= 1e-8 * np.linalg.norm(Hu), (tag, p)


def test_helpers_do_not_change_results(oracle_port, hjob_mode):
    A, B = [], []
    for i, (n, ir) in enumerate([(5000, 0.4), (1200, 0.3), (3000, 0.5), (400, 0.6), (2500, 0.2), (800, 0.4)]):
        p1, p2, _, _ = syn.homography_pairs(n, ir, 0.5, seed=230 + i, laf=True); A.append(p1); B.append(p2)
    seeds = [41 + i for i in range(len(A))]
    for err, et in (("sampson", 0), ("symm_sq_sum", 3)):
        ora = [oracle_port.find_homography(A[p], B[p], 2.0, 0.999, 50000, et, True, 3.0, seed=seeds[p]) for p in range(len(A))]
        assert sum(o[2]["lo_runs"] >= 2 for o in ora) >= 3
        for variant in (512, 256, 128):
            for place in (_lib.TUNE_PLACE_HBM, _lib.TUNE_PLACE_POOL_LDS, _lib.TUNE_PLACE_LDS):      # LDS: no helpers (points not in the workspace), same path otherwise
                for mode in (1, 0):
                    _lib.set_hjob_mode(mode)
                    H, m = pd.findHomographyBatch(A, B, 2.0, 0.999, 50000, 3.0, err, True, seeds=seeds, tuning=VARIANT[variant] | place)
                    _check(ora, H, m, pd.last_stats(), (err, variant, place, mode))


def test_a_batch_larger_than_the_device_gets_helpers_at_its_end(oracle_port, hjob_mode):
    rng = np.random.default_rng(11)
    A, B = [], []
    for i in range(1300):
        n = int(rng.choice([300, 1000, 2500])); p1, p2, _, _ = syn.homography_pairs(n, float(rng.uniform(0.2, 0.6)), 0.5, seed=5000 + i, laf=True); A.append(p1); B.append(p2)
    seeds = [int(x) for x in rng.integers(1, 2**31 - 1, len(A))]
    _lib.set_hjob_mode(0)
    H0, m0 = pd.findHomographyBatch(A, B, 2.0, 0.999, 20000, 3.0, "sampson", True, seeds=seeds); s0 = pd.last_stats()
    _lib.set_hjob_mode(1)
    H1, m1 = pd.findHomographyBatch(A, B, 2.0, 0.999, 20000, 3.0, "sampson", True, seeds=seeds); s1 = pd.last_stats()
    key = lambda st: [(x["samples"], x["lo_runs"], x["models"], x["rejected"], x["I"], x["best_sample"]) for x in st]
    assert key(s0) == key(s1)
    assert np.array_equal(np.asarray(H0), np.asarray(H1)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(m0, m1))
    order = np.argsort([-x["lo_runs"] for x in s1])[:16]
    for p in order:
        Ho, mo, so = oracle_port.find_homography(A[p], B[p], 2.0, 0.999, 20000, 0, True, 3.0, seed=seeds[p])
        assert (s1[p]["samples"], s1[p]["lo_runs"], s1[p]["models"]) == (so["samples"], so["lo_runs"], so["models"]), p
        assert np.array_equal(np.asarray(m1[p]), mo), p
