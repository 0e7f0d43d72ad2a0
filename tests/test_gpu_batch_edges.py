"""GPU: batch-level edge cases of the C-ABI — ragged batches that straddle the LDS limit, the size-independent
properties of a full-size batch, error codes."""
import ctypes as C

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu


def test_ragged_batch_across_the_lds_limit_equals_single_calls():
    """n = 64 .. 6000 in one launch: n_max decides the placement (pool in LDS, points in the workspace) for every pair."""
    sizes = [64, 2500, 6000, 8, 1000]
    A, B = [], []
    for i, n in enumerate(sizes):
        p1, p2, _, _ = syn.two_view_fundamental(n, 0.5, 0.1, seed=40 + i); A.append(p1); B.append(p2)
    seeds = [9, 8, 7, 6, 5]
    Fb, mb = pd.findFundamentalMatrixBatch(A, B, max_iters=5000, seeds=seeds)
    for p in range(len(sizes)):
        F1, m1 = pd.findFundamentalMatrix(A[p], B[p], max_iters=5000, seed=seeds[p])
        assert np.array_equal(np.asarray(Fb[p]), np.asarray(F1))
        assert np.array_equal(np.asarray(mb[p], dtype=bool), np.asarray(m1, dtype=bool))


def test_full_size_batch_properties():
    """1100 C2 pairs (enough for the throughput variant): results do not depend on the batch a pair sits in or on its
    position; every mask is consistent with its model; the stats block is sane."""
    P, N = 1100, 2000
    base = [syn.two_view_fundamental(N, 0.4, 0.1, seed=i)[:2] for i in range(20)]
    A = [base[i % 20][0] for i in range(P)]; B = [base[i % 20][1] for i in range(P)]
    seeds = [1 + (i % 20) for i in range(P)]                     # pair i == pair i + 20: identical problems
    F, m = pd.findFundamentalMatrixBatch(A, B, seeds=seeds)
    assert all(s_["threads"] == 256 for s_ in pd.last_stats()), "a batch of >= 3 pairs per CU must take the throughput variant"
    F = np.asarray(F)
    for i in range(20, P):
        assert np.array_equal(F[i], F[i % 20]) and np.array_equal(np.asarray(m[i]), np.asarray(m[i % 20]))
    F20, m20 = pd.findFundamentalMatrixBatch(A[:20], B[:20], seeds=seeds[:20])          # latency variant
    assert np.array_equal(np.asarray(F20), F[:20])
    for i in range(20):
        assert np.array_equal(np.asarray(m20[i]), np.asarray(m[i]))
        # mask == Sampson error <= th^2 under the returned model, recomputed in numpy (allowing knife-edge points)
        p1, p2 = A[i], B[i]; Fi = F[i]
        x1 = np.c_[p1, np.ones(N)]; x2 = np.c_[p2, np.ones(N)]
        Fx1 = x1 @ Fi.T; Ftx2 = x2 @ Fi
        d = (np.sum(x2 * Fx1, 1) ** 2) / (Fx1[:, 0] ** 2 + Fx1[:, 1] ** 2 + Ftx2[:, 0] ** 2 + Ftx2[:, 1] ** 2)
        mi = np.asarray(m[i], dtype=bool)
        assert (d[mi] <= 0.25 * (1 + 1e-9)).all() and mi.sum() >= 700


def test_error_codes():
    L = _lib.lib()
    p1, p2, _, _ = syn.two_view_fundamental(100, 0.5, 0.1, seed=1)
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch([p1[:7]], [p2[:7]], seeds=[1])           # n < 8 (bindings.cpp:270)
    with pytest.raises(ValueError):
        pd.findHomographyBatch([p1[:3]], [p2[:3]], seeds=[1])                  # n < 4 (bindings.cpp:35)
    with pytest.raises(_lib.MiDegensacError):
        pd.findFundamentalMatrix(p1, p2, seed=1, device=63)                    # no such device: no CPU fallback


def test_tuning_fields_that_do_not_apply_are_rejected():
    """every tuning field has its own bits (include/mi_degensac.h); one that does not apply to the call is EINVAL, not a
    silently different meaning"""
    p1, p2, _, _ = syn.two_view_fundamental(200, 0.5, 0.1, seed=1)
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch([p1], [p2], seeds=[1], tuning=_lib.TUNE_H_SERIAL_LO)
    with pytest.raises(ValueError):
        pd.findHomographyBatch([p1], [p2], seeds=[1], tuning=_lib.TUNE_COOP_ALL_PASSES)
    with pytest.raises(ValueError):
        pd.findHomographyBatch([p1], [p2], seeds=[1], tuning=_lib.TUNE_LONG_SHIFT(2))
    with pytest.raises(ValueError):
        pd.findHomographyBatch([p1], [p2], seeds=[1], tuning=_lib.TUNE_F_SERIAL_REPS)
    pd.findFundamentalMatrixBatch([p1], [p2], seeds=[1], tuning=_lib.TUNE_LONG_SHIFT(2) | _lib.TUNE_SET_ASIDE(4))      # applies: accepted
    pd.findHomographyBatch([p1], [p2], seeds=[1], tuning=_lib.TUNE_H_SERIAL_LO)


def test_bench_under_torchrun_with_the_rccl_process_group():
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run, backend "nccl" = RCCL), with one rank
    on this box's one GPU and --dist-always: process-group init on the device, the barrier, all_gather_into_tensor of the
    packed per-pair results on device tensors and the rank-0 JSON line all execute on hardware.  (No multi-GPU box is
    available to this suite; the world-size-2 logic runs under gloo in tests/test_host_cpu.py.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--pairs-per-gpu", "96", "--no-cpu-baseline", "--no-secondary", "--parity-pairs", "4", "--dist-always"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["parity_checked"] >= 4
    assert j["config"]["collective"].startswith("nccl")


def test_two_ranks_over_rccl_give_the_one_rank_results(tmp_path):
    """Runs the moment a box shows two devices: bench.py under torch.distributed.run with TWO ranks (backend nccl = RCCL over
    xGMI), 64 pairs per GPU, against the same 128 global pairs on one rank.  Pair ids, seeds and therefore every per-pair
    result (model, mask, counters) must not depend on the shard count; the process group must really be nccl with
    world size 2.  Skipped on the one-GPU boxes this suite normally gets."""
    import json, os, subprocess, sys
    import numpy as np, torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = []
    for world, ppg, port_no in ((1, 128, 29551), (2, 64, 29552)):
        dump = str(tmp_path / f"w{world}.npz")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port_no), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
               "--pairs-per-gpu", str(ppg), "--no-cpu-baseline", "--no-secondary", "--parity-pairs", "4", "--dist-always", "--dump-results", dump]
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert j["n_gpus"] == world and j["config"]["pairs_total"] == 128
        assert j["config"]["process_group"] == {"backend": "nccl", "world_size": world}
        outs.append(np.load(dump))
    for k in ("models", "masks"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert np.array_equal(outs[0]["stats"][:, :12], outs[1]["stats"][:, :12])        # words 12.. are device-clock times / variant
