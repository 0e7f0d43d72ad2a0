"""GPU: round 6 — every hand-over wait is bounded and fault-injectable (homography helpers, cooperative large-n mode), ransacH2el's
no-model mask equals the oracle's, the versioned diagnostics struct, per-call timing of the host-pointer path, bench.py's launch modes, the fan mode,
per-context scheduling, the homography kernel at two workgroups per CU."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_homography_helper_time_out_is_discarded_and_rerun(oracle_port):
    """Fault injection on the homography path (wait limit 0): the owner's wait for the repetitions its helper workgroups claimed fails at
    once (dg_wait_count), the launch raises its error word, the pairs are discarded and run again without helpers — the host-pointer
    entry point still returns the oracle's results (stats bit 11)."""
    sets = [syn.homography_pairs(3000, 0.4, 0.5, seed=40 + i, laf=True)[:2] for i in range(3)]
    A = [s_[0] for s_ in sets]; B = [s_[1] for s_ in sets]; seeds = [3, 4, 5]
    ora = [oracle_port.find_homography(A[p], B[p], 2.0, 0.999, 20000, 0, True, 3.0, seed=seeds[p]) for p in range(3)]
    prev = _lib.set_wait_ticks(0)
    try:
        H, k = pd.findHomographyBatch(A, B, 2.0, 0.999, 20000, 3.0, seeds=seeds, tuning=_lib.TUNE_LATENCY | _lib.TUNE_PLACE_POOL_LDS); st = pd.last_stats()
        assert sum(s_["rerun"] for s_ in st) >= 1, "with helpers on, the zero limit must have tripped a wait"
        assert sum(s_["discarded"] for s_ in st) == 0
        for p, (Ho, mo, so) in enumerate(ora):
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"]) == (so["samples"], so["lo_runs"], so["I"]), p
            assert np.array_equal(np.asarray(k[p]), mo), p
    finally:
        _lib.set_wait_ticks(prev)
    H2, k2 = pd.findHomographyBatch(A, B, 2.0, 0.999, 20000, 3.0, seeds=seeds, tuning=_lib.TUNE_LATENCY | _lib.TUNE_PLACE_POOL_LDS); st2 = pd.last_stats()
    assert sum(s_["rerun"] + s_["discarded"] for s_ in st2) == 0                    # with the default limit nothing trips
    assert np.array_equal(np.asarray(H), np.asarray(H2)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(k, k2))


def test_cooperative_mode_time_out_is_discarded_and_rerun(oracle_port):
    """The same for the cooperative large-n mode (fundamental matrix, 9000 correspondences, helper workgroups forced on): the owner's
    waits for claimed units give up at once, the pair is discarded and run again with the helpers off."""
    p1, p2, _, _ = syn.two_view_fundamental(9000, 0.3, 0.1, seed=77)
    Fo, mo, so = oracle_port.find_fundamental(p1, p2, 0.5, 0.9999, 3000, seed=9)
    tun = _lib.TUNE_LATENCY | _lib.TUNE_PLACE_HBM | _lib.TUNE_HELPERS(3)
    prev = _lib.set_wait_ticks(0)
    try:
        F, m = pd.findFundamentalMatrixBatch([p1], [p2], max_iters=3000, seeds=[9], tuning=tun); st = pd.last_stats()
        assert st[0]["rerun"] == 1 and st[0]["discarded"] == 0
        assert (st[0]["samples"], st[0]["lo_runs"], st[0]["I"]) == (so["samples"], so["lo_runs"], so["I"])
        assert np.array_equal(np.asarray(m[0]), mo)
        assert np.linalg.norm(np.asarray(F[0]).ravel() - Fo.ravel()) <= 1e-9 * np.linalg.norm(Fo)
    finally:
        _lib.set_wait_ticks(prev)
    F2, m2 = pd.findFundamentalMatrixBatch([p1], [p2], max_iters=3000, seeds=[9], tuning=tun); st2 = pd.last_stats()
    assert st2[0]["rerun"] == 0 and st2[0]["discarded"] == 0 and np.array_equal(np.asarray(m2[0]), mo)


def test_h2el_no_model_mask_equals_the_oracle(oracle_port):
    """ransacH2el when NOTHING ever becomes the best model (a threshold nothing meets, budgets of one or a few samples, with and without
    the run after the loop): errs[3] is the buffer as allocated.  Oracle and device take a zero-filled one: the raw mask
    is (0 <= th) for every point, the model stays zero — compared unconditionally, mask included — and the user-facing call turns
    the zero model into an all-false mask (findHomography's convention, utils.py:104-107)."""
    u10, _ = syn.ellipse_pairs(300, 0.0, 1.0, 301, 0.05)
    n_zero = 0
    for th, iters, lo in ((1e-12, 1, False), (1e-12, 1, True), (1e-12, 7, False), (1e-12, 60, True), (4.0, 1, False), (4.0, 3, True)):
        for seed in (1, 2, 3):
            Hr, mr = pd.ransacH2el(u10, th, 0.99, iters, lo, 0, seed=seed, raw=True); st = pd.last_stats()
            Ho, mo, so = oracle_port.ransacH2el(u10, th, 0.99, iters, lo, 0, seed)
            assert (st["samples"], st["lo_runs"], st["I"]) == (so["samples"], so["lo_runs"], so["I"]), (th, iters, lo, seed)
            assert np.array_equal(np.asarray(mr).astype(bool), mo), (th, iters, lo, seed, int(np.asarray(mr).sum()), int(mo.sum()))
            assert np.array_equal(np.asarray(Hr).ravel() == 0, Ho.ravel() == 0), (th, iters, lo, seed)
            if not Ho.any():
                n_zero += 1
                H, m = pd.ransacH2el(u10, th, 0.99, iters, lo, 0, seed=seed)
                assert not np.asarray(H).any() and not np.asarray(m).any()
    assert n_zero >= 3, "the sweep must contain runs that end without a model"


def test_diag_struct_is_versioned():
    """mi_degensac_diag.struct_size: a caller built against the first layout passes 0 in that field (it was `reserved`) and a struct
    that ends at d_hist — the library must not read d_screen then (here: a poisoned pointer that would fault if it were written through)."""
    import torch
    dev = torch.device("cuda", 0)
    p1, p2, _, _ = syn.two_view_fundamental(600, 0.5, 0.1, seed=2)
    offs = np.array([0, 600], np.int64)
    d_a = torch.from_numpy(p1).to(dev); d_b = torch.from_numpy(p2).to(dev); d_off = torch.from_numpy(offs).to(dev)
    d_seeds = torch.tensor([5], dtype=torch.int32, device=dev)
    prm = _lib.make_params(0.5, 0.9999, 5000, 0, True, 0.0, True)
    L = _lib.lib(); stream = torch.cuda.current_stream(dev)
    outs = []
    for size, scr in ((0, 0xdead0000), (None, None)):
        d_F = torch.zeros((1, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(600, dtype=torch.uint8, device=dev); d_st = torch.zeros((1, 16), dtype=torch.int32, device=dev)
        d_scr = torch.zeros((1, 4), dtype=torch.int32, device=dev)
        diag = _lib.Diag(None, 0, 0, None, scr if scr is not None else d_scr.data_ptr())
        if size is not None:
            diag.struct_size = size
        _lib.check(L.mi_degensac_find_fundamental_batch_dev_ex(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), 1, 2,
                                                               C.byref(prm), d_seeds.data_ptr(), 0, C.c_void_p(stream.cuda_stream), d_F.data_ptr(), d_mask.data_ptr(),
                                                               d_st.data_ptr(), C.byref(diag)))
        torch.cuda.synchronize(dev)
        outs.append((d_F.cpu().numpy(), d_mask.cpu().numpy(), d_scr.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert not outs[0][2].any() and outs[1][2][0, 3] > 0          # only the caller that states its size gets the screening counters


def test_call_timing_adds_up():
    """mi_degensac_set_call_timing / _last_call_timing: the phases of one host-pointer call sum to the call, the kernel lies inside the wait."""
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=0)
    prev = _lib.set_call_timing(1)
    try:
        pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, 100000, 0, True, 0.0, True, seed=1)
        pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, 100000, 0, True, 0.0, True, seed=2)
        t = _lib.last_call_timing()
    finally:
        _lib.set_call_timing(prev)
    parts = t["pack_ms"] + t["enqueue_ms"] + t["wait_ms"] + t["unpack_ms"]
    assert t["call_ms"] > 0 and abs(parts - t["call_ms"]) < 0.2 + 0.05 * t["call_ms"], t
    assert 0 < t["dev_kernel_ms"] <= t["call_ms"], t
    assert t["dev_h2d_ms"] >= 0 and t["dev_d2h_ms"] >= 0


def test_bench_gpus_flag_is_a_request_not_a_label():
    """bench.py --gpus 2 on a box with fewer GPUs must fail loudly instead of printing a one-GPU line; --single-process runs the
    device-list mode and reports the devices it used; a launcher whose WORLD_SIZE disagrees with --gpus is an error."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n_vis = torch.cuda.device_count()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_vis + 1), "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "visible" in out.stderr and not out.stdout.strip(), (out.returncode, out.stderr[-400:])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "must agree" in out.stderr
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-process", "--pairs-per-gpu", "64", "--steps", "1", "--warmup", "1",
                          "--parity-pairs", "3"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-800:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["process_group"] is None and line["parity_checked"] >= 3 and line["value"] > 0


def test_fan_mode_equals_the_cooperative_mode_and_the_oracle(oracle_port):
    """Fan mode (dg_f_fan.h; automatic for one large pair per owner on an idle device): the owner draws the sample stream into the ring, worker
    workgroups solve and score the chunks, the owner commits the completed entries in order.  Same results as the cooperative mode alone
    (MI_DEGENSAC_FLAG_NO_STREAM switches the workers off) and as the oracle, for two pairs at once, a budget that ends inside a chunk, a
    symmetric-epipolar metric and a plane-dominated scene (DEGENSAC branch: a bound that can fall); a worker that never answers
    (wait limit 0) ends in discard + re-run, not in wrong numbers."""
    cases = [dict(n=9000, ir=0.3, iters=6000, et="sampson", plane=0.0), dict(n=12000, ir=0.15, iters=10001, et="sampson", plane=0.0),
             dict(n=8500, ir=0.35, iters=4000, et="symm_epipolar", plane=0.0), dict(n=9000, ir=0.4, iters=5000, et="sampson", plane=0.7),
             dict(n=9000, ir=0.5, iters=100, et="sampson", plane=0.0), dict(n=9000, ir=0.5, iters=300, et="sampson", plane=0.0),       # budgets of one and two chunks
             dict(n=9000, ir=0.5, iters=513, et="sampson", plane=0.0)]
    for cs in cases:
        A, B = [], []
        for i in range(2):
            p1, p2, _, _ = syn.two_view_fundamental(cs["n"], cs["ir"], 0.1, seed=900 + i, plane_fraction=cs["plane"]); A.append(p1); B.append(p2)
        seeds = [5, 6]
        tun = _lib.TUNE_LATENCY | _lib.TUNE_PLACE_HBM          # (below ~19 000 correspondences the sampler pool would stay in LDS: no cooperative mode, no fan)
        F, m = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, cs["iters"], error_type=cs["et"], seeds=seeds, tuning=tun); st = pd.last_stats()
        F0, m0 = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, cs["iters"], error_type=cs["et"], seeds=seeds, tuning=tun, flags=_lib.FLAG_NO_STREAM); s0 = pd.last_stats()
        assert sum(s_["streamed"] for s_ in st) == 2 and sum(s_["streamed"] for s_ in s0) == 0, cs        # stats bit 9: the chunks came from the ring
        key = lambda s: [(x["samples"], x["lo_runs"], x["models"], x["degen"], x["I"], x["best_sample"]) for x in s]
        assert np.array_equal(np.asarray(F), np.asarray(F0)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(m, m0)) and key(st) == key(s0), cs
        et = 1 if cs["et"] == "symm_epipolar" else 0
        for p in range(2):
            Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, cs["iters"], et, True, 0.0, True, seed=seeds[p])
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"], st[p]["degen"]) == (so["samples"], so["lo_runs"], so["I"], so["degen"]), (cs, p)
            assert np.array_equal(np.asarray(m[p]), mo), (cs, p)
            assert np.linalg.norm(np.asarray(F[p]).ravel() - Fo.ravel()) <= 1e-9 * np.linalg.norm(Fo), (cs, p)
    prev = _lib.set_wait_ticks(0)
    try:
        p1, p2, _, _ = syn.two_view_fundamental(9000, 0.3, 0.1, seed=900)
        F, m = pd.findFundamentalMatrixBatch([p1], [p2], 0.5, 0.9999, 6000, seeds=[5], tuning=_lib.TUNE_LATENCY | _lib.TUNE_PLACE_HBM); st = pd.last_stats()
        Fo, mo, so = oracle_port.find_fundamental(p1, p2, 0.5, 0.9999, 6000, seed=5)
        assert st[0]["discarded"] == 0 and st[0]["rerun"] == 1 and np.array_equal(np.asarray(m[0]), mo) and st[0]["samples"] == so["samples"]
    finally:
        _lib.set_wait_ticks(prev)


def test_scheduling_defaults_belong_to_a_context():
    """mi_degensac_ctx_set_scheduling: the stream mode / helper choice of ONE context; other contexts and the process-wide defaults are
    untouched, per-call flags still win, results are the same either way."""
    L = _lib.lib()
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=0)
    off = np.array([0, 2000], np.int64); seeds = np.array([3], np.uint32)

    def run(ctx, flags=0):
        F = np.zeros(9); m = np.zeros(2000, np.uint8); st = np.zeros(16, np.int32)
        prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, flags)
        _lib.check(L.mi_degensac_ctx_find_fundamental_batch(ctx, _lib.dptr(p1), _lib.dptr(p2), off.ctypes.data_as(C.POINTER(C.c_int64)), 1, 2, C.byref(prm),
                                                            seeds.ctypes.data_as(C.POINTER(C.c_uint32)), _lib.dptr(F), m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                            st.ctypes.data_as(C.POINTER(C.c_int32))))
        return F, m, (int(st[15]) >> 9) & 1, tuple(int(x) for x in st[:8])
    a = C.c_void_p(); b = C.c_void_p()
    _lib.check(L.mi_degensac_ctx_create(0, C.byref(a))); _lib.check(L.mi_degensac_ctx_create(0, C.byref(b)))
    try:
        assert L.mi_degensac_ctx_set_scheduling(a, 0, -2) == 0                       # context a: stream mode off
        assert L.mi_degensac_ctx_set_scheduling(b, 1 | (2 << 1), -2) == 0            # context b: on, pairs ask at once
        Fa, ma, sa, ka = run(a); Fb, mb, sb, kb = run(b)
        assert sa == 0 and sb == 1
        assert np.array_equal(Fa, Fb) and np.array_equal(ma, mb) and ka == kb
        assert run(a, _lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2))[2] == 1 and run(b, _lib.FLAG_NO_STREAM)[2] == 0     # the call's own flags win
        assert L.mi_degensac_ctx_set_scheduling(a, -3, 0) != 0
    finally:
        L.mi_degensac_ctx_destroy(a); L.mi_degensac_ctx_destroy(b)


def test_homography_batches_between_one_and_two_pairs_per_cu_run_two_workgroups_per_cu(oracle_port):
    """The 256-thread homography kernel fits two workgroups per CU (its last wave's solver table lives in LDS members only the
    fundamental-matrix kernel uses) and is the host's choice for batches of more than one and at most two pairs per CU: same results as the
    512-thread kernel, and the oracle's."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    P = cus + 9
    sets = [syn.homography_pairs(300 + 7 * (i % 5), 0.5, 0.5, seed=900 + i, laf=True)[:2] for i in range(P)]
    A = [s_[0] for s_ in sets]; B = [s_[1] for s_ in sets]; seeds = [11 + i for i in range(P)]
    H, k = pd.findHomographyBatch(A, B, 2.0, 0.999, 5000, 3.0, seeds=seeds); st = pd.last_stats()
    assert {s_["threads"] for s_ in st} == {256}, {s_["threads"] for s_ in st}
    H2, k2 = pd.findHomographyBatch(A, B, 2.0, 0.999, 5000, 3.0, seeds=seeds, tuning=_lib.TUNE_LATENCY); st2 = pd.last_stats()
    assert {s_["threads"] for s_ in st2} == {512}
    assert np.array_equal(np.asarray(H), np.asarray(H2)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(k, k2))
    for p in (0, 1, P // 2, P - 1):
        Ho, mo, so = oracle_port.find_homography(A[p], B[p], 2.0, 0.999, 5000, 0, True, 3.0, seed=seeds[p])
        assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"]) == (so["samples"], so["lo_runs"], so["I"]), p
        assert np.array_equal(np.asarray(k[p]), mo), p


def test_fundamental_batches_of_more_than_two_pairs_per_cu_take_the_256_thread_kernel(oracle_port):
    """The variant crossover of the fundamental-matrix kernel: up to two pairs per CU the 512-thread kernel (with producers), past that the
    256-thread kernel at two workgroups per CU; results do not depend on the choice."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    base = [syn.two_view_fundamental(200 + 9 * i, 0.5, 0.1, seed=700 + i)[:2] for i in range(8)]
    for P, want in ((2 * cus, 512), (2 * cus + 8, 256)):
        A = [base[i % 8][0] for i in range(P)]; B = [base[i % 8][1] for i in range(P)]; seeds = [21 + i for i in range(P)]
        F, k = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.999, 3000, seeds=seeds); st = pd.last_stats()
        assert {s_["threads"] for s_ in st} == {want}, (P, {s_["threads"] for s_ in st})
        F2, k2 = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.999, 3000, seeds=seeds, tuning=_lib.TUNE_THROUGHPUT4); st2 = pd.last_stats()
        assert np.array_equal(np.asarray(F), np.asarray(F2)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(k, k2))
        for p in (0, P - 1):
            Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.999, 3000, 0, True, 0.0, True, seed=seeds[p])
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"]) == (so["samples"], so["lo_runs"], so["I"]), p
            assert np.array_equal(np.asarray(k[p]), mo), p
