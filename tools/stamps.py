#!/usr/bin/env python3
"""tools/stamps.py [bench args]: per-phase cycle totals of the strip kernels (library built by
`python tools/build_variant.py stamps -DRH_STAMPS`); prints average cycles per workgroup and phase."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACTIP_HOT_LIB"] = os.path.join(ROOT, "ractip_amd", "libractip_hot_stamps.so")
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
names = ["stage loads+lds writes", "barrier", "fm2 pre-phase", "chain operand loads", "filter", "barrier", "partials+barrier", "chain", "", "", "", "", "", "", ""]
print("workgroups:", wg)
for k in range(15):
    if buf[k]:
        print("%-28s %10.0f cycles/workgroup (100 MHz ticks x ?)" % (names[k], buf[k] / wg))
