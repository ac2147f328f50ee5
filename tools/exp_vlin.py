#!/usr/bin/env python3
"""Development check: Vienna-BL linear path (mode 2) vs the CPU restatement."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ractip_amd
from _oracle import ViennaOracle
from ractip_amd.seqgen import random_pairs
vo = ViennaOracle()
c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
c.set_mode(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
rng = np.random.RandomState(4)
seqs = ["GGGAAACCCAGGGAAACCCA", "GGAAACCCAGGGAAACCC", "GGCGAAAGCCAAGGCGAAAGCCAA", "CUCGGCUUGCUGAGGUGCACACAGCAAGAGGCGAG", "GAAAC", "A"]
seqs += ["".join(rng.choice(list("ACGU"), n)) for n in (17, 63, 64, 65, 130, 200, 331)]
for s in seqs:
    o = vo.mccaskill(s, max_w=4)
    bp, z = c.bpp(s)
    big = o["post"] > 1e-12
    rel = (np.abs(bp - o["post"])[big] / o["post"][big]).max() if big.any() else 0.0
    print("n=%4d logZ %.9f (ref %.9f) d=%.2e  bp max rel %.2e  abs small %.2e  path %d" % (
        len(s), z, o["logZ"], abs(z - o["logZ"]), rel, np.abs(bp - o["post"])[~big].max() if (~big).any() else 0, c.last_path()))
    if len(sys.argv) > 2:
        up = c.unpaired(s, max_w=4)
        bigu = o["up"] > 1e-11
        print("      up max rel %.2e" % ((np.abs(up - o["up"])[bigu] / o["up"][bigu]).max() if bigu.any() else 0.0))
