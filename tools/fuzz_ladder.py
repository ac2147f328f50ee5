#!/usr/bin/env python3
"""tools/fuzz_ladder.py [seconds] [seed]: random streams of batches that mix ordinary sequences with chains of stable hairpins (outside the
double range of the default scale exponent), both models, ragged lengths, on ONE context per model (so that the remembered exponent
moves back and forth); every batch is compared with a context that has the ladder switched off (log-space fallback)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ractip_amd
from _oracle import assert_prob_close

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
comp = {"G": "C", "C": "G"}

def hairpins(n, k):
    s = ""
    while len(s) < n:
        stem = "".join(rng.choice(list("GC"), size=k))
        s += stem + "AAAA" + "".join(comp[ch] for ch in reversed(stem)) + "A" * int(rng.integers(1, 5))
    return s[:n]

def rnd(n):
    return "".join(rng.choice(list("ACGU"), size=n))

V = ractip_amd.hot.RH_MODEL_VIENNA_BL
ladder = {"cf": ractip_amd.Context(device=0), "vi": ractip_amd.Context(device=0, model=V)}
os.environ["RH_SCALE_LADDER"] = "0"
plain = {"cf": ractip_amd.Context(device=0), "vi": ractip_amd.Context(device=0, model=V)}
del os.environ["RH_SCALE_LADDER"]
for c in (ladder["vi"], plain["vi"]):
    c.set_hybrid(True)
for c in ladder.values():
    c.set_scale_memory(True)   # the remembered exponent is what this tool exercises (off by default)
t0, batches, rescaled_total, moved = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    model = "cf" if rng.random() < 0.6 else "vi"
    npairs = int(rng.integers(1, 9))
    frac = rng.choice([0.0, 0.2, 0.6, 1.0])
    top = 1500 if model == "cf" else 1000
    def one():
        n = int(rng.integers(40, top))
        return hairpins(n, int(rng.integers(7, 12))) if rng.random() < frac else rnd(n)
    pairs = [(one(), one()) for _ in range(npairs)]
    out = []
    for c in (ladder[model], plain[model]):
        c.batch_upload(pairs); c.batch_compute()
        out.append([c.batch_results(p) for p in range(npairs)])
    rescaled_total += len(ladder[model].batch_fallbacks(2))
    for p, (r, r0) in enumerate(zip(*out)):
        tag = "%s batch %d pair %d (%d/%d)" % (model, batches, p, len(pairs[p][0]), len(pairs[p][1]))
        assert np.allclose(r["logZ"], r0["logZ"], rtol=1e-8, atol=1e-8), (tag, r["logZ"], r0["logZ"])
        for k in ("bp1", "bp2", "hp"):
            assert_prob_close(r[k], r0[k], rel=1e-6, what=tag + " " + k)
        for k in ("up1", "up2"):
            assert np.abs(np.asarray(r[k]) - np.asarray(r0[k])).max() < 1e-8, (tag, k)
    batches += 1
    if batches % 20 == 0:
        print("%d batches, %d rescaled sequences, %.0f s" % (batches, rescaled_total, time.time() - t0), flush=True)
print("ladder fuzz ok: %d batches, %d rescaled sequences in %.0f s" % (batches, rescaled_total, time.time() - t0))
