"""CPU suite: host logic of the drop-in layer, the C-ABI library's exported symbols, and the
pair-sharding / result gather over a world_size-2 gloo group."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from pydegensac_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    hdr = open(os.path.join(ROOT, "include", "mi_degensac.h")).read()
    names = sorted(set(re.findall(r"\b(mi_degensac_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 12
    lib = C.CDLL(_lib.LIB_PATH)          # loads without a GPU; no compute is called
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mi_degensac.h but not exported"
    lib.mi_degensac_version.restype = C.c_char_p
    assert b"gfx950" in lib.mi_degensac_version()


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import pydegensac_amd as pd
    from pydegensac_amd import _lib
    p = np.random.default_rng(0).uniform(0, 100, (50, 2))
    with pytest.raises(_lib.MiDegensacError):
        pd.findFundamentalMatrix(p, p + 1.0, seed=1)


def test_python_surface_matches_reference_signatures():
    import inspect
    import pydegensac_amd as pd
    s = inspect.signature(pd.findHomography)
    assert list(s.parameters)[:8] == ["pts1_", "pts2_", "px_th", "conf", "max_iters", "laf_consistensy_coef", "error_type",
                                      "symmetric_error_check"]
    assert [s.parameters[k].default for k in list(s.parameters)[2:8]] == [1.0, 0.999, 50000, -1.0, "sampson", True]
    s = inspect.signature(pd.findFundamentalMatrix)
    assert list(s.parameters)[:9] == ["pts1_", "pts2_", "px_th", "conf", "max_iters", "laf_consistensy_coef", "error_type",
                                      "symmetric_error_check", "enable_degeneracy_check"]
    assert [s.parameters[k].default for k in list(s.parameters)[2:9]] == [0.5, 0.9999, 100000, -1.0, "sampson", True, True]
    assert pd.api.error_type_dict_homography == {"sampson": 0, "symm_sq_max": 1, "symm_max": 2, "symm_sq_sum": 3, "symm_sum": 4}
    assert pd.api.error_type_dict_fundamental == {"sampson": 0, "symm_epipolar": 1}


def test_input_validation_errors_like_reference():
    import pydegensac_amd as pd
    good = np.zeros((10, 2))
    with pytest.raises(ValueError):
        pd.findHomography(np.zeros((10, 3)), good)          # utils.py:50-52
    with pytest.raises(ValueError):
        pd.findHomography(np.zeros((3, 2)), np.zeros((3, 2)))   # utils.py:53-54
    with pytest.raises(ValueError):
        pd.findHomography("nope", good)                      # utils.py:68-70
    with pytest.raises(AssertionError):
        pd.findHomography(np.zeros((10, 2)), np.zeros((11, 2)))  # utils.py:86
    with pytest.raises(ValueError):
        pd.findHomography(good, good, error_type="bogus")    # utils.py:93
    with pytest.raises(ValueError):
        pd.findFundamentalMatrix(good, good, error_type="symm_max")
    with pytest.raises(ValueError):
        pd.findFundamentalMatrix(np.zeros((5, 2)), np.zeros((5, 2)))   # bindings.cpp:270-272 (N >= 8)


def test_shard_ranges_cover_everything():
    from pydegensac_amd import parallel
    for n in [1, 7, 8, 4096, 4099]:
        for w in [1, 2, 3, 8]:
            r = [parallel.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
    assert parallel.pair_seed(5) == parallel.pair_seeds(5, 6)[0]


GLOO_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pydegensac_amd import parallel, synthetic
from oracle import port            # the CPU oracle stands in for the GPU kernel in this CPU-only test
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
NP, N = 5, 200
lo, hi = parallel.shard_range(NP, rank, world)
F = np.zeros((hi - lo, 9)); st = np.zeros((hi - lo, 16), np.int32); mk = np.zeros(((hi - lo) * N,), np.uint8)
for i, pid in enumerate(range(lo, hi)):
    p1, p2, _, _ = synthetic.two_view_fundamental(N, 0.6, 0.1, seed=pid)
    f, m, s = port.find_fundamental(p1, p2, max_iters=2000, seed=parallel.pair_seed(pid))
    F[i] = f.ravel(); mk[i * N:(i + 1) * N] = m; st[i, 0] = s["samples"]; st[i, 3] = s["I"]
gm, gs, gk = parallel.gather_results(torch.from_numpy(F), torch.from_numpy(st), torch.from_numpy(mk), N, NP)
# ragged batch: pair p has 100 + 37 p correspondences
sizes = [100 + 37 * p for p in range(NP)]
Fr = np.zeros((hi - lo, 9)); sr = np.zeros((hi - lo, 16), np.int32); mr = []
for i, pid in enumerate(range(lo, hi)):
    p1, p2, _, _ = synthetic.two_view_fundamental(sizes[pid], 0.6, 0.1, seed=50 + pid)
    f, m, s = port.find_fundamental(p1, p2, max_iters=1000, seed=parallel.pair_seed(pid))
    Fr[i] = f.ravel(); mr.append(m.astype(np.uint8)); sr[i, 0] = s["samples"]
rm, rs, rk = parallel.gather_results(torch.from_numpy(Fr), torch.from_numpy(sr), torch.from_numpy(np.concatenate(mr)), sizes, NP)
# very uneven shards (one 5000-correspondence pair next to tiny ones): every rank ships its own bytes, results in pair order
big = [5000, 10, 11, 12, 13]
rngb = np.random.default_rng(7)
allm = [rngb.integers(0, 2, size=b).astype(np.uint8) for b in big]; allF = rngb.normal(size=(NP, 9)); alls = rngb.integers(0, 1000, size=(NP, 16)).astype(np.int32)
bm, bs, bk = parallel.gather_results(torch.from_numpy(allF[lo:hi].copy()), torch.from_numpy(alls[lo:hi].copy()),
                                     torch.from_numpy(np.concatenate(allm[lo:hi])), big, NP)
assert np.array_equal(bm.numpy(), allF) and np.array_equal(bs.numpy(), alls) and np.array_equal(bk.numpy(), np.concatenate(allm))
# shards with EQUAL packed record sizes but different pair counts (world 2: 2 x 100 and 1 x 336 are both 472 bytes): must not
# take the one-collective path, whose slices assume this rank's pair count for every rank
eq = [100, 100, 336]
alle = [rngb.integers(0, 2, size=b).astype(np.uint8) for b in eq]; eF = rngb.normal(size=(3, 9)); es = rngb.integers(0, 1000, size=(3, 16)).astype(np.int32)
elo, ehi = parallel.shard_range(3, rank, world)
em, est, ek = parallel.gather_results(torch.from_numpy(eF[elo:ehi].copy()), torch.from_numpy(es[elo:ehi].copy()),
                                      torch.from_numpy(np.concatenate(alle[elo:ehi]) if ehi > elo else np.zeros(0, np.uint8)), eq, 3)
assert em.shape == (3, 9) and np.array_equal(em.numpy(), eF) and np.array_equal(est.numpy(), es) and np.array_equal(ek.numpy(), np.concatenate(alle))
if rank == 0:
    assert rk.numel() == sum(sizes)
    np.savez(sys.argv[2], F=gm.numpy(), st=gs.numpy(), mk=gk.numpy(), Fr=rm.numpy(), sr=rs.numpy(), mr=rk.numpy())
dist.destroy_process_group()
'''


def test_gather_invariant_to_world_size(tmp_path):
    from oracle import port
    port.lib()
    script = tmp_path / "w.py"; script.write_text(GLOO_WORKER)
    outs = []
    for world in [1, 2, 3]:
        out = str(tmp_path / f"o{world}.npz")
        port_no = 29500 + os.getpid() % 1000 + world
        if world == 1:
            env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no))
            subprocess.check_call([sys.executable, str(script), ROOT, out], env=env)
        else:
            subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                                   "--master-addr", "127.0.0.1", "--master-port", str(port_no), str(script), ROOT, out])
        outs.append(np.load(out))
    for k in ["F", "st", "mk", "Fr", "sr", "mr"]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert np.array_equal(outs[0][k], outs[2][k]), k
    assert outs[0]["st"][:, 0].min() > 0


def test_tensor_api_rejects_host_and_wrong_dtype_inputs():
    """tensor_api validates before touching the library: CPU tensors / float32 / shape mismatches are ValueErrors."""
    import torch
    from pydegensac_amd import tensor_api
    a = torch.zeros((16, 2), dtype=torch.float64)
    with pytest.raises(ValueError):
        tensor_api.find_fundamental_batch_tensors(a, a, [16])                     # not on a GPU
    with pytest.raises(ValueError):
        tensor_api.find_fundamental_batch_tensors(a.numpy(), a.numpy(), [16])     # not tensors


def test_pydegensac_alias_runs_the_reference_example_call_pattern():
    """`import pydegensac` resolves to this implementation with the reference's surface (src/pydegensac/__init__.py:1-4);
    the call pattern of examples/simple-example.py:18-37 (float32 [n,2] arrays, positional th / conf / n_iter) goes
    through unchanged — up to the device: without a GPU the call must fail loudly, never fall back."""
    import inspect
    import torch
    import pydegensac
    import pydegensac_amd
    from pydegensac_amd import _lib
    for name in ["findHomography", "findFundamentalMatrix", "convert_cv2_kpts_to_xyA", "findHomography_", "findFundamentalMatrix_"]:
        assert getattr(pydegensac, name) is getattr(pydegensac_amd, name)
    s = inspect.signature(pydegensac.findFundamentalMatrix_)
    assert [s.parameters[k].default for k in ["px_th", "conf", "max_iters"]] == [0.5, 0.9999, 200000]     # bindings.cpp:494-503
    s = inspect.signature(pydegensac.findHomography_)
    assert [s.parameters[k].default for k in ["px_th", "conf", "max_iters"]] == [1.0, 0.999, 10000]       # bindings.cpp:484-492
    from pydegensac_amd import synthetic
    p1, p2, lab, _ = synthetic.homography_pairs(n=300, inlier_ratio=0.6, sigma=0.5, seed=1)
    src_pts = np.float32(p1).reshape(-1, 2); dst_pts = np.float32(p2).reshape(-1, 2)
    if torch.cuda.is_available():
        H, mask = pydegensac.findHomography(src_pts, dst_pts, 4.0, 0.99, 2000)                   # simple-example.py:21
        F, maskf = pydegensac.findFundamentalMatrix(src_pts, dst_pts, 1.0, 0.999, 10000, enable_degeneracy_check=True)   # :35
        assert H.shape == (3, 3) and int(np.asarray(mask).astype(np.float32).sum()) > 100
    else:
        with pytest.raises(_lib.MiDegensacError):
            pydegensac.findHomography(src_pts, dst_pts, 4.0, 0.99, 2000)


def test_batch_api_validates_every_pair():
    import pydegensac_amd as pd
    a = np.zeros((20, 2)); b = np.zeros((20, 2))
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch([a, a], [b, b[:19]], seeds=[1, 2])          # shorter pts2 in pair 1
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch([a, np.zeros((20, 6))], [b, np.zeros((20, 6))], seeds=[1, 2])   # mixed dims
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch([a], [b], error_type="bogus", seeds=[1])    # ValueError like the single-pair API
    with pytest.raises(ValueError):
        pd.findHomographyBatch([np.zeros((20, 3))], [np.zeros((20, 3))], seeds=[1])
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch([a, a], [b, b], seeds=[1])                  # one seed per pair


def test_alias_package_has_the_reference_submodules():
    """the reference ships `pydegensac/utils.py` and the extension module `pydegensac.pydegensac` (src/pydegensac/__init__.py:1-4):
    both import paths must work on the alias package"""
    import importlib
    u = importlib.import_module("pydegensac.utils")
    ext = importlib.import_module("pydegensac.pydegensac")
    from pydegensac.utils import convert_and_check, findHomography, findFundamentalMatrix, convert_cv2_kpts_to_xyA   # noqa: F401
    import pydegensac_amd as pd
    assert u.findHomography is pd.findHomography and ext.findFundamentalMatrix_ is pd.findFundamentalMatrix_
    a = convert_and_check(np.arange(12).reshape(6, 2))
    assert a.dtype == np.float64 and a.shape == (6, 2)


def test_ellipse_ransac_validates_its_input_before_touching_the_device():
    """ransacH2el (ranH2el.h:35): u10 must be [n, 10] with n >= 2, one seed per pair — checked on the host"""
    import pydegensac_amd as pd
    from pydegensac_amd import synthetic as syn
    u, lab = syn.ellipse_pairs(40, 0.5, 1.0, 3, 0.05)
    assert u.shape == (40, 10) and lab.sum() == 20
    with pytest.raises(ValueError):
        pd.ransacH2el(u[:, :6])
    with pytest.raises(ValueError):
        pd.ransacH2el(u[:1])
    with pytest.raises(ValueError):
        pd.ransacH2el_batch([])
    with pytest.raises(ValueError):
        pd.ransacH2el_batch([u, u], seeds=[1])


def test_cv2_keypoint_list_branch_with_a_stub_cv2_module():
    """utils.py:57-66: a list of cv2.KeyPoint goes through convert_cv2_kpts_to_xyA; a list of anything else is a ValueError.
    OpenCV is absent from this image, so the branch runs against a stub `cv2` module that only has the KeyPoint type (pt, size,
    angle — everything the reference's converter reads, utils.py:24-41); the module is reloaded so its `import cv2` sees the stub."""
    import importlib
    import types
    import pydegensac_amd.api as api

    class KeyPoint:
        def __init__(self, x, y, size, angle):
            self.pt = (x, y); self.size = size; self.angle = angle

    stub = types.ModuleType("cv2"); stub.KeyPoint = KeyPoint
    had = sys.modules.get("cv2")
    sys.modules["cv2"] = stub
    try:
        importlib.reload(api)
        assert api.OPENCV_HERE
        kps = [KeyPoint(10.0 + i, 20.0 - i, 4.0 + 0.5 * i, 30.0 * i) for i in range(6)]
        out = api.convert_and_check(kps)
        assert out.shape == (6, 6) and out.dtype == np.float64
        for i, kp in enumerate(kps):
            a = np.deg2rad(kp.angle); s = kp.size
            assert np.allclose(out[i], [kp.pt[0], kp.pt[1], s * np.cos(a), s * np.sin(a), -s * np.sin(a), s * np.cos(a)], atol=1e-12)
        with pytest.raises(ValueError):
            api.convert_and_check([(1.0, 2.0)] * 6)               # a list, but not of cv2.KeyPoint (utils.py:58-61)
        with pytest.raises(ValueError):
            api.convert_and_check((1, 2, 3))                      # neither an array nor a list (utils.py:68-70)
    finally:
        if had is None:
            sys.modules.pop("cv2", None)
        else:
            sys.modules["cv2"] = had
        importlib.reload(api)
    assert not api.OPENCV_HERE or had is not None
    with pytest.raises(ValueError):
        api.convert_and_check([object()] * 6)                     # without OpenCV: "Cannot import cv2" (utils.py:66)


def test_bench_gpus_request_resolution():
    """bench.py --gpus N (round-5 review: the flag was parsed and ignored).  N > 1 without a launcher re-executes under
    torch.distributed.run with N ranks on 127.0.0.1; fewer visible GPUs than requested, or a launcher whose WORLD_SIZE disagrees, is an
    error — never a one-GPU line labelled otherwise."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert bench.resolve_world(1, False, {}, 1) == ("run", 1)
    assert bench.resolve_world(8, False, {}, 8) == ("exec", 8)
    assert bench.resolve_world(4, True, {}, 8) == ("run", 1)                      # --single-process: this process drives the 4 devices
    assert bench.resolve_world(8, False, {"WORLD_SIZE": "8", "LOCAL_RANK": "7"}, 8) == ("run", 8)
    for args in ((2, False, {}, 1), (8, False, {}, 0), (1, False, {"WORLD_SIZE": "2"}, 8), (8, False, {"WORLD_SIZE": "4"}, 8),
                 (2, True, {"WORLD_SIZE": "2"}, 2), (0, False, {}, 1), (2, False, {"WORLD_SIZE": "2", "LOCAL_RANK": "1"}, 1)):
        with pytest.raises(SystemExit) as e:
            bench.resolve_world(*args)
        assert e.value.code not in (0, None) and "bench.py" in str(e.value.code)
    cmd = bench.relaunch_argv(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], port=29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    port_cmd = bench.relaunch_argv(2, [])
    assert 1024 < int(port_cmd[port_cmd.index("--master-port") + 1]) < 65536
    # the whole script, no GPU here: a request for two GPUs exits non-zero with the reason, and prints no JSON line
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode != 0 and "only 0 GPU(s) are visible" in out.stderr and not out.stdout.strip()


def test_bench_model_distance_is_up_to_sign():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rng = np.random.default_rng(1); a = rng.normal(size=9)
    d, flip = bench.model_rel(a, 3.0 * a); assert d < 1e-15 and not flip
    d, flip = bench.model_rel(a, -0.5 * a); assert d < 1e-15 and flip
    d, flip = bench.model_rel(a, a + 1e-3 * rng.normal(size=9)); assert 1e-5 < d < 1e-2 and not flip
    assert bench.model_rel(np.zeros(9), np.zeros(9)) == (0.0, False) and bench.model_rel(np.zeros(9), a)[0] == 1.0
