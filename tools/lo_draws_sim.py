"""CPU only: how predictable is the number of generator draws a repetition of the fundamental-matrix local optimisation consumes?

The one-repetition-per-wave order (dg_inFrani_waves) starts the repetitions of a round from generator states that ASSUME a draw count
for every earlier repetition of the round; a round commits repetitions up to and including the first one whose count differs.  This
script takes the per-repetition draw counts of the CPU oracle (trace tags 10 / 15 / 11 / 13 / 14 / 12 of exp_iterFcustom) on C2 pairs and
counts the rounds several assumption rules would need with 2 / 4 / 8 waves.   usage: python tools/lo_draws_sim.py [pairs]"""
import sys, os, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import port
from pydegensac_amd import synthetic as syn, parallel

P = int(sys.argv[1]) if len(sys.argv) > 1 else 48
L = port.lib()
CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_double)
ev = []
cb = CB(lambda tag, I, J: ev.append((tag, I)))
runs = []                                   # list of LO runs, each a list of per-repetition draw counts
for p in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=p)
    ev.clear(); L.dg_oracle_set_trace2(cb)
    port.find_fundamental(p1, p2, 0.5, 0.9999, 100000, seed=int(parallel.pair_seed(p)))
    L.dg_oracle_set_trace2(CB(0))
    cur = None; reps = None
    for tag, I in ev:
        if tag == 2:                         # a LO run starts (the LSQ before it): exp_ranF.c:1511
            if reps: runs.append(reps)
            reps = []
        elif tag == 10 and reps is not None: reps.append(0)
        elif tag == 15 and reps: reps[-1] += 8 if I > 8 else 0
        elif tag == 14 and reps: reps[-1] += 8 if I > 8 else 0
    if reps: runs.append(reps)
runs = [r for r in runs if len(r) == 10]
allc = collections.Counter(d for r in runs for d in r)
print("LO runs", len(runs), "draw counts", dict(sorted(allc.items())))
print("by repetition index:", [dict(sorted(collections.Counter(r[i] for r in runs).items())) for i in range(10)])

def rounds(run, nw, rule):
    """rule(q, history_of_committed_counts) -> assumed count of repetition q"""
    nxt = 0; n = 0; hist = []
    while nxt < len(run):
        n += 1
        k = 0
        while k < nw and nxt + k < len(run):
            d = run[nxt + k]; a = rule(nxt + k, hist)
            hist.append(d); k += 1
            if d != a: break
        nxt += k
    return n
rules = {
    "always 16": lambda q, h: 16,
    "last committed": lambda q, h: h[-1] if h else 16,
    "last, one odd count ignored": None,
    "16, first repetition 24": lambda q, h: 24 if q == 0 else 16,
    "most frequent so far in this run": lambda q, h: collections.Counter(h).most_common(1)[0][0] if h else 16,
}
def sticky():
    st = {"a": 16, "p": -1}
    def f(q, h):
        if q == 0: st["a"], st["p"] = 16, -1
        if h:
            d = h[-1]
            if d == st["p"] or st["p"] < 0: st["a"] = d
            st["p"] = d
        return st["a"]
    return f
rules["last, one odd count ignored"] = sticky()
for nw in (2, 4, 8):
    print("waves", nw, {name: round(sum(rounds(r, nw, (lambda q, h, f=f: f(q, h[len(h) - q:] if q else []))) for r in runs) / len(runs), 2) for name, f in rules.items()},
          "ideal", -(-10 // nw))
