"""GPU: the BASELINE configurations at their STATED sizes against goldens of the unmodified reference
(tests/golden/make_golden.py), and the timed bench batch / the device-resident API against the oracle.

  C5   findFundamentalMatrix, 50 000 correspondences, 10 % inliers, max_iters 200 000, conf 0.9999
  C3   findHomography, 5 000 correspondences WITH LAFs, LAF + symmetric checks, sampson and symm_max
  C2b  dominant plane, all 100 000 samples (reference quirk: exp_ranF.c:1571-1576 only updates max_sam inside an LO)
"""
import os

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, parallel, synthetic as syn
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def _gold(name):
    paths = [p for p in gu.fixtures(name[0]) if os.path.basename(p).startswith(name)]
    assert len(paths) == 2, name
    return paths


def _run_f(g, tuning=0):
    kw = g["call"]
    F, m = pd.findFundamentalMatrix_(g["p1"], g["p2"], kw.get("px_th", 0.5), kw.get("conf", 0.9999), kw.get("max_iters", 100000),
                                     kw.get("error_type", 0), kw.get("sym_check", True), kw.get("laf_coef", 0.0),
                                     kw.get("degen", True), seed=g["seed"], tuning=tuning)
    return F, m, pd.last_stats()


@pytest.mark.parametrize("path", _gold("F_c5_full"), ids=lambda p: os.path.basename(p)[:-4])
def test_c5_full_size_matches_reference_golden(path):
    g = gu.load(path)
    assert g["n"] == 50000 and g["call"]["max_iters"] == 200000
    F, m, st = _run_f(g)
    assert st["samples"] == g["samples"] and st["lo_runs"] == g["lo_runs"]
    assert st["full_passes"] == g["full_passes"] and st["ex_passes"] == g["ex_passes"]
    assert np.array_equal(np.asarray(m), g["mask"]), f"{(np.asarray(m) != g['mask']).sum()} mask bits differ"
    assert gu.rel(F, g["model"]) < 1e-6


@pytest.mark.parametrize("path", _gold("F_c2b_full"), ids=lambda p: os.path.basename(p)[:-4])
def test_c2b_all_100000_samples_matches_reference_golden(path):
    g = gu.load(path)
    F, m, st = _run_f(g)
    assert g["samples"] == 100000, "this fixture is meant to run the full length (SURVEY 3.4 #1)"
    assert st["samples"] == g["samples"] and st["lo_runs"] == g["lo_runs"]
    assert st["full_passes"] == g["full_passes"] and st["ex_passes"] == g["ex_passes"]
    assert np.array_equal(np.asarray(m), g["mask"])
    assert gu.rel(F, g["model"]) < 1e-6


@pytest.mark.parametrize("name", ["H_c3_full_laf_sampson", "H_c3_full_laf_symm_max"])
@pytest.mark.parametrize("s", [0, 1])
def test_c3_laf_at_5000_matches_reference_golden(name, s):
    g = gu.load(_gold(name)[s]); kw = g["call"]
    assert g["n"] == 5000 and g["p1"].shape[1] == 6 and kw["laf_coef"] == 3.0
    H, m = pd.findHomography_(g["p1"], g["p2"], kw["px_th"], kw.get("conf", 0.999), kw.get("max_iters", 50000),
                              kw["error_type"], kw.get("sym_check", True), kw["laf_coef"], seed=g["seed"])
    st = pd.last_stats()
    assert (st["samples"], st["lo_runs"], st["rejected"]) == (g["samples"], g["lo_runs"], g["rejected"])
    assert st["models"] == g["full_passes"]
    assert np.array_equal(np.asarray(m), g["mask"])
    assert gu.rel(H, g["model"]) < 1e-6


def _bench_batch(P):
    """the first P pairs of bench.py's timed batch: data seed = pair id, RANSAC seed = parallel.pair_seed(pair id)"""
    A = []; B = []
    for pid in range(P):
        p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=pid); A.append(p1); B.append(p2)
    return A, B, parallel.pair_seeds(0, P)


def test_throughput_variant_batch_of_1100_against_oracle(oracle_port):
    """1100 pairs of the bench workload through the 256-thread variant (what the timed 4096-pair batch runs), 64 randomly
    chosen pairs compared with the oracle: masks, models, sample / LO / scored-model counters."""
    P = 1100
    A, B, seeds = _bench_batch(P)
    F, m = pd.findFundamentalMatrixBatch(A, B, seeds=seeds)
    st = pd.last_stats()
    assert all(s_["threads"] == 256 for s_ in st)
    pick = [int(x) for x in np.random.default_rng(11).choice(P, 64, replace=False)]
    # the batch queues behind the resident grid, so pairs are set aside after the discovery round and resumed by priority:
    # both kinds must be among the checked pairs (pairs that finish inside the discovery round are never set aside)
    aside = [p for p in range(P) if st[p]["set_aside"]]
    assert len(aside) >= 8, len(aside)
    pick = sorted(set(pick + aside[:: max(1, len(aside) // 8)][:8]))
    assert sum(st[p]["set_aside"] for p in pick) >= 8 and any(not st[p]["set_aside"] for p in pick)
    for p in pick:
        Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 100000, seed=int(seeds[p]))
        assert (st[p]["samples"], st[p]["lo_runs"], st[p]["degen"]) == (so["samples"], so["lo_runs"], so["degen"]), p
        assert st[p]["full_passes"] == so["full_passes"] and st[p]["ex_passes"] == so["ex_passes"], p
        assert np.array_equal(np.asarray(m[p]), mo), p
        assert gu.rel(F[p], Fo) < 1e-6, p


def test_c3_batch_of_1024_takes_the_small_workgroups_and_matches_the_oracle(oracle_port):
    """The bench's C3 batch (1024 homography pairs, 5000 correspondences with LAFs): the host picks 128-thread workgroups
    with the sampler pool in the workspace (four resident pairs per CU); 24 randomly chosen pairs against the oracle, and
    the set-aside machinery of the F driver stays off."""
    P = 1024
    A, B = [], []
    for i in range(P):
        p1, p2, _, _ = syn.homography_pairs(5000, 0.4, 0.5, seed=i, laf=True); A.append(p1); B.append(p2)
    seeds = [int(x) for x in parallel.pair_seeds(0, P)]
    H, m = pd.findHomographyBatch(A, B, 2.0, 0.999, 50000, 3.0, "sampson", True, seeds=seeds)
    st = pd.last_stats()
    assert all(s_["threads"] == 128 and s_["placement"] == 0 and s_["set_aside"] == 0 for s_ in st), st[0]
    for p in np.random.default_rng(5).choice(P, 24, replace=False):
        Ho, mo, so = oracle_port.find_homography(A[p], B[p], 2.0, 0.999, 50000, 0, True, 3.0, seed=seeds[p])
        assert (st[p]["samples"], st[p]["lo_runs"], st[p]["rejected"]) == (so["samples"], so["lo_runs"], so["rejected"]), p
        assert np.array_equal(np.asarray(m[p]), mo), p
        Hu = np.linalg.inv(Ho.T)                                    # utils.py:108
        assert np.linalg.norm(np.asarray(H[p]) - Hu) <= 1e-6 * np.linalg.norm(Hu), p


def test_tensor_api_against_oracle(oracle_port):
    """device-resident API: a ragged F batch and an H batch with LAFs, every pair against the oracle"""
    import torch
    from pydegensac_amd import tensor_api
    dev = torch.device("cuda", 0)
    sizes = [700, 1200, 64, 2000, 333, 1999]
    A = []; B = []
    for i, n in enumerate(sizes):
        p1, p2, _, _ = syn.two_view_fundamental(n, 0.4, 0.1, seed=60 + i); A.append(p1); B.append(p2)
    seeds = [3, 5, 7, 11, 4000000000, 13]                          # incl. a seed >= 2^31
    F, m, st, offs = tensor_api.find_fundamental_batch_tensors(torch.from_numpy(np.concatenate(A)).to(dev),
                                                               torch.from_numpy(np.concatenate(B)).to(dev), sizes,
                                                               max_iters=20000, seeds=seeds)
    F = F.cpu().numpy(); m = m.cpu().numpy(); st = st.cpu().numpy()
    for p in range(len(sizes)):
        Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=seeds[p])
        assert st[p, 0] == so["samples"] and st[p, 1] == so["lo_runs"], p
        assert np.array_equal(m[offs[p]:offs[p + 1]], mo), p
        assert gu.rel(F[p], Fo) < 1e-6, p
    sizes = [900, 300, 1500]
    A = []; B = []
    for i, n in enumerate(sizes):
        p1, p2, _, _ = syn.homography_pairs(n, 0.4, 0.5, seed=70 + i, laf=True); A.append(p1); B.append(p2)
    seeds = [2, 4, 6]
    H, m, st, offs = tensor_api.find_homography_batch_tensors(torch.from_numpy(np.concatenate(A)).to(dev),
                                                              torch.from_numpy(np.concatenate(B)).to(dev), sizes, 1.0, 0.999, 20000,
                                                              3.0, "symm_max", True, seeds=seeds)
    H = H.cpu().numpy(); m = m.cpu().numpy(); st = st.cpu().numpy()
    for p in range(len(sizes)):
        Ho, mo, so = oracle_port.find_homography(A[p], B[p], 1.0, 0.999, 20000, 2, True, 3.0, seed=seeds[p])
        assert (st[p, 0], st[p, 1], st[p, 2]) == (so["samples"], so["lo_runs"], so["rejected"]), p
        assert np.array_equal(m[offs[p]:offs[p + 1]], mo), p
        Hu = np.linalg.inv(Ho.T)                                    # utils.py:108
        assert np.linalg.norm(H[p] - Hu) <= 1e-6 * np.linalg.norm(Hu), p
