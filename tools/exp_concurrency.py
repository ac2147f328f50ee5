#!/usr/bin/env python3
"""Experiment: do independent contexts (own streams/graphs) overlap on one GPU?  N contexts x B pairs each, all computing
concurrently from N host threads, vs one context with N*B pairs."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ractip_amd
from ractip_amd.seqgen import random_pairs

def run(nctx, batch, n=500, steps=8):
    pairs = random_pairs(batch * nctx, n, seed=12345)
    ctxs = [ractip_amd.Context(device=0) for _ in range(nctx)]
    for k, c in enumerate(ctxs):
        c.batch_upload(pairs[k * batch:(k + 1) * batch])
        c.batch_compute()
    def work(c):
        for _ in range(steps):
            c.batch_compute()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    for c in ctxs: c.close()
    return nctx * batch * steps / dt

for nctx, batch in ((1, 256), (2, 128), (4, 64), (2, 256), (1, 512), (3, 96)):
    print("contexts %d x %3d pairs: %8.1f pairs/s" % (nctx, batch, run(nctx, batch)), flush=True)
