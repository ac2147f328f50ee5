#!/usr/bin/env python3
"""tools/exp_small.py: the one-workgroup-per-sequence kernel (mccaskill_small.hip, RH_SMALL=1) against the sweeps on short sequences --
agreement of bp / log Z per length, and the time of 1000 pairs of 109 + 53 letters under both."""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import ractip_amd
from ractip_amd.seqgen import random_pairs
rng = np.random.RandomState(3)
def rnd(n): return "".join("ACGU"[k] for k in rng.randint(0, 4, n))
seqs = [rnd(n) for n in (8, 9, 30, 53, 64, 65, 66, 100, 109, 110, 111, 112)]
def run(env):
    os.environ["RH_SMALL"] = env
    c = ractip_amd.Context(device=0)
    out = [c.bpp(s) for s in seqs]
    pairs = [(rnd(109), rnd(53)) for _ in range(1000)]
    c.batch_upload(pairs); c.batch_compute()
    t = time.time()
    for _ in range(5): c.batch_compute()
    dt = (time.time() - t) / 5
    tm = c.batch_timings()
    c.close()
    return out, dt, tm
a, ta, tma = run("1")
b, tb, tmb = run("0")
for s, (bp1, z1), (bp0, z0) in zip(seqs, a, b):
    print(len(s), "dz=%.3e" % abs(z1 - z0), "max|dbp|=%.3e" % np.abs(bp1 - bp0).max(), "identical" if np.array_equal(bp1, bp0) else "")
print("1000 pairs (109,53): small %.3f ms  sweeps %.3f ms" % (ta * 1e3, tb * 1e3), tma, tmb)
