"""Matcher stage throughput: brute-force 2-NN over n1 x n2 descriptors on device-resident tensors.
usage: gpu_matcher.py [n1 n2]   -> time per call (HIP events), direct-form flop/s (3 n1 n2 dim for L2), pairs/s"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import tensor_api
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
for name, a, b, dim in (("float32 x 128 (L2)", torch.randn(n1, 128, generator=g), torch.randn(n2, 128, generator=g), 128),
                        ("uint8 x 32 (Hamming)", torch.randint(0, 256, (n1, 32), generator=g, dtype=torch.uint8), torch.randint(0, 256, (n2, 32), generator=g, dtype=torch.uint8), 32)):
    a = a.to(dev); b = b.to(dev)
    for _ in range(2): tensor_api.knn_match_tensors(a, b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): idx, dist = tensor_api.knn_match_tensors(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ops = 3.0 * n1 * n2 * dim if a.dtype == torch.float32 else 2.0 * n1 * n2 * (dim // 4)     # sub, mul, add per word / xor + popcount per 32-bit word
    print(f"{name}: {n1} x {n2}: {ms:.3f} ms per call, {ops / ms / 1e9:.2f} T{'flop' if a.dtype == torch.float32 else 'op'}/s, {n1 * n2 / ms / 1e6:.1f} G pair-distances/s")
for nq in (1500,):
    a = torch.randn(nq, 128, generator=g).to(dev); b = torch.randn(nq, 128, generator=g).to(dev)
    for _ in range(2): tensor_api.knn_match_tensors(a, b)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): tensor_api.knn_match_tensors(a, b)
    e1.record(); torch.cuda.synchronize()
    print(f"example-sized call ({nq} x {nq} x 128 float32): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
