#!/usr/bin/env python3
"""tools/vienna_isolated.py [pairs]: one Vienna-BL batch (n = 500, hp from the two-molecule ensemble) computed a few times with the
streams NOT overlapping -- run it under `rocprofv3 --kernel-trace --stats` to read per-kernel times without contention."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ractip_amd
from ractip_amd.seqgen import random_pairs
pairs = random_pairs(int(sys.argv[1]) if len(sys.argv) > 1 else 128, 500, seed=12345)
c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
c.set_hybrid(True)
c.set_overlap(False)
c.batch_upload(pairs)
for _ in range(4):
    c.batch_compute()
print(c.batch_timings())
c.close()
