"""GPU: stream mode of the fundamental-matrix kernel (dg_stream_cb).  A pair with many samples left hands its sample
stream, 7-point solves and screening / scoring to a PRODUCER workgroup (one that has no pair of its own) and only commits the
chunks in order.  Results must not depend on it: every run below is compared with the CPU oracle pair by pair, with the mode
off, on, on with the owner re-scoring every chunk (the path taken when the bound falls after a DEGENSAC completion) and on
while unstarted pairs remain."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu

VARIANT = {512: _lib.TUNE_LATENCY, 256: _lib.TUNE_THROUGHPUT, 128: _lib.TUNE_THROUGHPUT4}
PLACE = {1: _lib.TUNE_PLACE_LDS, 2: _lib.TUNE_PLACE_POOL_LDS}


@pytest.fixture()
def stream_mode():
    prev = _lib.set_stream_mode(-1)
    yield
    _lib.set_stream_mode(prev)


def _batch():
    A, B = [], []
    # plane-dominated scenes keep their whole sample budget (DEGENSAC branch), the others end early: both kinds side by side
    for i, (n, pf, ir) in enumerate([(1500, 0.7, 0.4), (2000, 0.0, 0.4), (400, 0.9, 0.5), (900, 0.0, 0.25), (2000, 0.6, 0.4), (800, 0.8, 0.3),
                                     (300, 0.0, 0.15), (1200, 0.5, 0.6)]):
        p1, p2, _, _ = syn.two_view_fundamental(n, ir, 0.1, seed=170 + i, plane_fraction=pf); A.append(p1); B.append(p2)
    return A, B, [31 + i for i in range(len(A))]


def _check(ora, F, m, st, tag):
    for p, (Fo, mo, so) in enumerate(ora):
        key = lambda s_: (s_["samples"], s_["lo_runs"], s_["degen"], s_["Ih"], s_["models"], s_["best_sample"])
        assert key(st[p]) == key(so), (tag, p, st[p], so)
        assert np.array_equal(np.asarray(m[p]), mo.astype(bool)), (tag, p)
        a = np.asarray(F[p]).ravel(); b = np.asarray(Fo).ravel()
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b), (tag, p)


def test_producer_workgroups_do_not_change_results(oracle_port, stream_mode):
    A, B, seeds = _batch()
    ora = [oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 30000, seed=seeds[p]) for p in range(len(A))]
    assert sum(o[2]["samples"] == 30000 for o in ora) >= 3, "some pairs must keep their whole budget"
    for variant in (512, 256, 128):
        for place in (1, 2):
            for mode in (0, 1, 3, 5, 7):        # off / on / on + re-score every chunk / on + ask at once / both
                _lib.set_stream_mode(mode)
                F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=30000, seeds=seeds, tuning=VARIANT[variant] | PLACE[place])
                st = pd.last_stats()
                _check(ora, F, m, st, (variant, place, mode))
                streamed = sum(s_["streamed"] for s_ in st)
                assert (streamed == 0) if mode == 0 else (streamed >= 3), (variant, place, mode, streamed)


def test_single_calls_get_a_producer(oracle_port, stream_mode):
    """one pair per call (the reference's use case): the launch carries a second workgroup that becomes the pair's producer"""
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=2, plane_fraction=0.0)
    for seed in (3, 4, 5):
        Fo, mo, so = oracle_port.find_fundamental(p1, p2, 0.5, 0.9999, 100000, seed=seed)
        for mode in (0, 1, 3):
            _lib.set_stream_mode(mode)
            F, m = pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, 100000, 0, True, 0.0, True, seed=seed)
            st = pd.last_stats()
            assert (st["samples"], st["lo_runs"], st["models"]) == (so["samples"], so["lo_runs"], so["models"]), (seed, mode)
            assert np.array_equal(np.asarray(m), mo), (seed, mode)
            assert np.linalg.norm(np.asarray(F).ravel() - np.asarray(Fo).ravel()) <= 1e-9 * np.linalg.norm(Fo), (seed, mode)


def test_a_batch_larger_than_the_device_streams_its_tail(oracle_port, stream_mode):
    """600 pairs on 256 CUs: tickets run out, the workgroups that finish first become producers of the pairs still running.
    Every pair against the same batch with the mode off; 24 pairs (the long ones first) against the oracle."""
    rng = np.random.default_rng(5)
    A, B = [], []
    for i in range(600):
        n = int(rng.choice([300, 800, 1500, 2000])); pf = float(rng.choice([0.0, 0.0, 0.0, 0.7]))
        p1, p2, _, _ = syn.two_view_fundamental(n, float(rng.uniform(0.25, 0.6)), 0.1, seed=3000 + i, plane_fraction=pf); A.append(p1); B.append(p2)
    seeds = [int(x) for x in rng.integers(1, 2**31 - 1, 600)]
    _lib.set_stream_mode(0)
    F0, m0 = pd.findFundamentalMatrixBatch(A, B, max_iters=40000, seeds=seeds); s0 = pd.last_stats()
    _lib.set_stream_mode(1)                       # on request: a batch of more than two pairs per resident workgroup leaves it off by itself
    F1, m1 = pd.findFundamentalMatrixBatch(A, B, max_iters=40000, seeds=seeds); s1 = pd.last_stats()
    assert sum(s_["streamed"] for s_ in s1) >= 5
    key = lambda st: [(x["samples"], x["lo_runs"], x["models"], x["degen"], x["I"], x["best_sample"]) for x in st]
    assert key(s0) == key(s1)
    assert np.array_equal(np.asarray(F0), np.asarray(F1)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(m0, m1))
    order = np.argsort([-x["samples"] for x in s1])[:24]
    for p in order:
        Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 40000, seed=seeds[p])
        assert (s1[p]["samples"], s1[p]["lo_runs"], s1[p]["models"]) == (so["samples"], so["lo_runs"], so["models"]), p
        assert np.array_equal(np.asarray(m1[p]), mo), p
