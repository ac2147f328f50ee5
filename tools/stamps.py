#!/usr/bin/env python3
"""tools/stamps.py [n [pairs]]: per-phase s_memtime totals of one strip kernel (library built by
`python tools/build_variant.py stamps -DRH_STAMPS=1` for the inside kernel, `=2` for the outside kernel); prints the
average ticks per workgroup between consecutive RH_STAMP sites (mccaskill_strip.hip), wavefront 0 of each workgroup."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACTIP_HOT_LIB"] = os.environ.get("RH_STAMPS_LIB") or os.path.join(ROOT, "ractip_amd", "libractip_hot_stamps.so")   # RH_STAMPS_LIB: another tuning build
import numpy as np
import ractip_amd
from ractip_amd import hot
lib = ctypes.CDLL(os.environ["RACTIP_HOT_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(1)
pairs = [("".join("ACGU"[k] for k in rng.integers(0, 4, n)), "".join("ACGU"[k] for k in rng.integers(0, 4, n))) for _ in range(npairs)]
ctx = ractip_amd.Context(device=0)
ctx.batch_upload(pairs)
ctx.batch_compute()
buf = (ctypes.c_ulonglong * 16)()
lib.rh_debug_stamps(buf, 1)
ctx.batch_compute()
lib.rh_debug_stamps(buf, 1)
wg = max(1, buf[15])
print("workgroups:", wg)
order = [0, 1, 2, 3, 9, 10, 11, 4, 5, 6, 7, 8, 12, 13, 14]
tot = sum(buf[k] for k in range(15)) / wg
for k in order:
    if buf[k]:
        print("stamp %2d  %10.0f ticks/workgroup  %5.1f %%" % (k, buf[k] / wg, 100.0 * buf[k] / wg / tot))
print("total     %10.0f" % tot)
