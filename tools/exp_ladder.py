#!/usr/bin/env python3
"""tools/exp_ladder.py [vienna]: a batch of 64 pairs of 1100-nt hairpin chains (log Z 0.58 per nucleotide: outside the double range with the
default scale exponent): time per batch with the scale-exponent ladder (rescaled on the linear kernels) and without (log-space kernels).
`vienna`: the Vienna-BL model (hp from the two-molecule ensemble), 63 ordinary pairs of 500 + one pair with a 900-nt hairpin chain."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and os.environ.get("RH_EXP_CHILD"):
    import numpy as np
    import ractip_amd
    rng = np.random.default_rng(5)
    comp = {"G": "C", "C": "G"}
    def hairpins(n):
        s = ""
        while len(s) < n:
            stem = "".join(rng.choice(list("GC"), size=10))
            s += stem + "AAAA" + "".join(comp[ch] for ch in reversed(stem)) + "AA"
        return s[:n]
    vienna = sys.argv[1] == "vienna"
    pairs = [(hairpins(1100), hairpins(1100)) for _ in range(64)]
    c = ractip_amd.Context(device=0)
    if vienna:
        from ractip_amd.seqgen import random_pairs
        pairs = random_pairs(63, 500, seed=3) + [(hairpins(900), pairs[0][1][:500])]
        c.close()
        c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
        c.set_hybrid(True)
    c.set_scale_memory(True)
    c.batch_upload(pairs); c.batch_compute()
    t0 = time.perf_counter()
    for _ in range(3):
        c.batch_upload(pairs); c.batch_compute()
    dt = (time.perf_counter() - t0) / 3
    print("RH_SCALE_LADDER=%s: %.1f ms per batch of 64 pairs = %.0f pairs/s; path %d, log-space %d, rescaled %d sequences"
          % (os.environ.get("RH_SCALE_LADDER", "1"), dt * 1e3, 64 / dt, c.last_path(), len(c.batch_fallbacks(0)), len(c.batch_fallbacks(2))))
    c.close()
else:
    for v in ("1", "0"):
        subprocess.check_call([sys.executable, __file__, sys.argv[1] if len(sys.argv) > 1 else "run"], env=dict(os.environ, RH_SCALE_LADDER=v, RH_EXP_CHILD="1"))
