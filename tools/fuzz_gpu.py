#!/usr/bin/env python3
"""Differential fuzzing on the GPU box: random inputs through the C ABI against the CPU oracles, both models, both
arithmetic paths, ragged batches, cuts at every offset of the 64-cell groups / 16-letter blocks, random constraints.
  python tools/fuzz_gpu.py [seconds] [seed] [length scale] [cf]      (cf: CONTRAfold pair batches only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ractip_amd
from _oracle import Oracle, ViennaOracle, assert_prob_close

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # multiplies the length ranges (3: up to 540 letters, many block-product tiles)
ONLY_CF = len(sys.argv) > 4 and sys.argv[4] == "cf"
cf, vo = Oracle(), ViennaOracle()
V = ractip_amd.hot.RH_MODEL_VIENNA_BL
ctxs = {("cf", m): ractip_amd.Context(device=0) for m in (0, 1)}
ctxs.update({("vi", m): ractip_amd.Context(device=0, model=V) for m in (0, 1)})
for (k, m), c in ctxs.items():
    c.set_mode(m)

def rnd(n, gc=0.5):
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    return "".join(rng.choice(list("ACGU"), n, p=p))

def rand_constraint(s):
    n = len(s)
    c = ["."] * n
    for _ in range(rng.randint(0, 4)):
        c[rng.randint(n)] = rng.choice(list("x<>|"))
    comp = {("A", "U"), ("U", "A"), ("G", "C"), ("C", "G"), ("G", "U"), ("U", "G")}
    for _ in range(rng.randint(0, 3)):     # nested forced pairs of complementary letters
        i, j = sorted(rng.randint(0, n, 2))
        if j - i >= 4 and (s[i], s[j]) in comp and all(ch == "." for ch in c[i:j + 1]):
            c[i], c[j] = "(", ")"
    return "".join(c)

t0, cases = time.time(), 0
while time.time() - t0 < budget:
    kind = 0 if ONLY_CF else rng.randint(5)
    m = rng.randint(2)
    gc = rng.choice([0.3, 0.5, 0.7])
    if kind == 0:      # CONTRAfold ragged pair batch
        pairs = [(rnd(rng.randint(1, 180 * SCALE), gc), rnd(rng.randint(1, 180 * SCALE), gc)) for _ in range(rng.randint(1, 5))]
        c = ctxs[("cf", m)]
        c.batch_upload(pairs); c.batch_compute()
        if m == 0:   # ordinary sequences must stay on the linear path: a fallback here would hide a kernel bug behind the log-space result
            assert c.last_path() == 1, ("silent McCaskill fallback", c.batch_fallbacks(0), pairs)
            # (the duplex of a ~1000-nt 70 % GC sequence can legitimately leave the double range at its fixed scale: per-pair fallback)
            assert c.last_hybrid_path() == 1 or gc > 0.6, ("silent duplex fallback", c.batch_fallbacks(1), pairs)
        for p, (s1, s2) in enumerate(pairs):
            r = c.batch_results(p)
            o1, o2, od = cf.inference(s1), cf.inference(s2), cf.duplex(s1, s2)
            assert_prob_close(r["bp1"], o1["post"], what="cf bp1 %r" % (s1,))
            assert_prob_close(r["bp2"], o2["post"], what="cf bp2 %r" % (s2,))
            assert_prob_close(r["hp"], od["post"], what="cf hp %r %r" % (s1, s2))
            assert abs(r["logZ"][0] - o1["logZ"]) < 1e-8 and abs(r["logZ"][2] - od["logZ2"][0]) < 1e-8
    elif kind == 1:    # Vienna ragged pair batch, both hp sources
        pairs = [(rnd(rng.randint(1, 150 * SCALE), gc), rnd(rng.randint(1, 150 * SCALE), gc)) for _ in range(rng.randint(1, 4))]
        co = bool(rng.randint(2))
        c = ctxs[("vi", m)]
        c.set_hybrid(co); c.set_max_w(int(rng.choice([1, 5, 15])))
        c.batch_upload(pairs); c.batch_compute()
        for p, (s1, s2) in enumerate(pairs):
            r = c.batch_results(p)
            o1 = vo.mccaskill(s1, max_w=c.max_w)
            oh = vo.cofold(s1, s2)["hp"] if co else vo.pf_duplex(s1, s2)["pr"]
            assert_prob_close(r["bp1"], o1["post"], what="vi bp1 %r" % (s1,))
            assert_prob_close(np.asarray(r["up1"]).reshape(len(s1), -1), o1["up"], abs_floor=1e-11, what="vi up1 %r w=%d" % (s1, c.max_w))
            assert_prob_close(r["hp"], oh, what="vi hp co=%s %r %r" % (co, s1, s2))
        c.set_hybrid(False); c.set_max_w(15)
    elif kind == 2:    # two-molecule ensemble: the cut at every offset relative to groups and blocks
        n1, n2 = rng.randint(1, 140 * SCALE), rng.randint(1, 140 * SCALE)
        s1, s2 = rnd(n1, gc), rnd(n2, gc)
        hp, z = ctxs[("vi", m)].cofold(s1, s2)
        o = vo.cofold(s1, s2)
        assert abs(z - o["logZ"]) < 1e-8 * max(1, abs(z)), (s1, s2)
        assert_prob_close(hp, o["hp"], what="cofold %r %r" % (s1, s2))
    elif kind == 3:    # constrained single fold
        s = rnd(rng.randint(5, 120 * SCALE), gc)
        cons = rand_constraint(s)
        o = vo.mccaskill(s, max_w=15, constraint=cons)
        bp, up, z = ctxs[("vi", m)].fold(s, constraint=cons)
        assert abs(z - o["logZ"]) < 1e-8 * max(1, abs(z)), (s, cons)
        assert_prob_close(bp, o["post"], what="constrained %r %r" % (s, cons))
        assert_prob_close(up, o["up"], abs_floor=1e-11, what="constrained up %r %r" % (s, cons))
    else:              # constrained two-molecule ensemble
        s1, s2 = rnd(rng.randint(3, 60), gc), rnd(rng.randint(3, 60), gc)
        cons = rand_constraint(s1 + s2)
        o = vo.cofold(s1, s2, constraint=cons)
        hp, z = ctxs[("vi", m)].cofold(s1, s2, constraint=cons)
        if np.isfinite(o["logZ"]):
            assert abs(z - o["logZ"]) < 1e-8 * max(1, abs(z)), (s1, s2, cons)
        assert_prob_close(hp, o["hp"], what="constrained cofold %r %r %r" % (s1, s2, cons))
    cases += 1
    if cases % 25 == 0:
        print("%d cases, %.0f s" % (cases, time.time() - t0), flush=True)
print("fuzz ok: %d cases in %.0f s" % (cases, time.time() - t0))
