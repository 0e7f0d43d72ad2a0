"""CPU: the parallel formulation of the sampler's Fisher-Yates pool stage (dg_sample_pool_par in
pydegensac_amd/csrc/dg_kernel_f_main.h), restated in numpy, equals the sequential reference (rtools.c:12-23: for draw i
of a sample, swap pool[s] with pool[n-1-i], the drawn id is what was at s).  The device code relies on one extra
hardware fact — lanes of one LDS atomic exchange that hit the same address are served in ascending lane order — which
tools/gpu_atomic_order.py probes on the GPU; here that order is simply the loop order."""
import numpy as np
import pytest


def sequential(pool, draws, n):
    pool = pool.copy(); ids = np.zeros_like(draws)
    for k in range(draws.shape[0]):
        for i in range(draws.shape[1]):
            s = draws[k, i]; j = n - 1 - i
            q = pool[s]; pool[s] = pool[j]; pool[j] = q; ids[k, i] = q
    return ids, pool


def parallel(pool0, draws, n):
    cn, nd = draws.shape; m2 = 2 * cn * nd
    vp = pool0.astype(np.int64).copy()                       # low 16 bits: id, upper bits: last toucher + 1
    ptr = np.zeros(m2, np.int64); pos = np.zeros(m2, np.int64)
    for u in range(m2):                                      # phase A: touches in swap order (64 per atomic instruction)
        tau, side = u >> 1, u & 1; k, i = divmod(tau, nd)
        p = draws[k, i] if side == 0 else n - 1 - i
        pos[u] = p
        old = vp[p]; vp[p] = (u + 1) << 16
        if old >> 16 == 0:
            ptr[u] = -1 - (old & 0xFFFF)                      # first touch: the id stored there
        else:
            v = (old >> 16) - 1                               # predecessor touch
            ptr[u] = u - 1 if (v >> 1) == tau else v ^ 1      # value = what the other side of that swap held
    rounds = 0
    while (ptr >= 0).any():                                   # phase B: pointer jumping, two hops per round
        nxt = ptr.copy()
        for x in np.nonzero(ptr >= 0)[0]:
            q = ptr[ptr[x]]
            nxt[x] = ptr[q] if q >= 0 else q
        ptr = nxt; rounds += 1
    ids = (-1 - ptr[0::2]).reshape(cn, nd)                    # phase C
    last = [(vp[pos[u]] >> 16) == u + 1 for u in range(m2)]   # phase D: last touchers store what their swap left
    for u in range(m2):
        if last[u]:
            vp[pos[u]] = -1 - ptr[u ^ 1]
    return ids, vp, rounds


@pytest.mark.parametrize("n,nd,cn", [(2000, 7, 256), (5000, 4, 256), (300, 7, 256), (64, 7, 100), (9, 7, 64), (8, 7, 256), (5, 4, 37),
                                     (1000, 4, 1), (40, 4, 256)])
def test_parallel_pool_stage_equals_sequential(n, nd, cn):
    rng = np.random.default_rng(n * 131 + nd * 17 + cn)
    pool = rng.permutation(n)
    for _ in range(3):                                        # consecutive chunks on the persisting pool
        draws = np.stack([rng.integers(0, n - i, size=cn) for i in range(nd)], 1)
        ids_s, pool_s = sequential(pool, draws, n)
        ids_p, pool_p, rounds = parallel(pool, draws, n)
        assert np.array_equal(ids_s, ids_p) and np.array_equal(pool_s, pool_p)
        assert sorted(pool_p.tolist()) == list(range(n))      # still a permutation, markers cleared
        assert rounds <= 12
        pool = pool_s
