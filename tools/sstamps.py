#!/usr/bin/env python3
"""tools/sstamps.py [n1 n2 pairs]: per-phase s_memtime totals of lin_small_fold (tuning build `python tools/build_variant.py sstamps -DRH_SMALL_STAMPS`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACTIP_HOT_LIB"] = os.environ.get("RH_STAMPS_LIB") or os.path.join(ROOT, "ractip_amd", "libractip_hot_sstamps.so")
import numpy as np
import ractip_amd
lib = ctypes.CDLL(os.environ["RACTIP_HOT_LIB"])
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 109
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 53
npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rng = np.random.default_rng(1)
pairs = [("".join("ACGU"[k] for k in rng.integers(0, 4, n1)), "".join("ACGU"[k] for k in rng.integers(0, 4, n2))) for _ in range(npairs)]
ctx = ractip_amd.Context(device=0)
ctx.batch_upload(pairs)
ctx.batch_compute()
buf = (ctypes.c_ulonglong * 16)()
lib.rh_debug_sstamps(buf, 1)
ctx.batch_compute()
lib.rh_debug_sstamps(buf, 1)
wg = max(1, buf[15])
names = ["inside: term loops", "inside: barrier", "inside: epilogue", "inside: second barrier", "F5 chains", "outside: term loops", "outside: barrier",
         "outside: epilogue", "outside: second barrier"]
tot = sum(buf[k] for k in range(9)) / wg
print("workgroups:", wg, ctx.batch_timings())
for k in range(9):
    print("%-26s %9.0f ticks/workgroup  %5.1f %%" % (names[k], buf[k] / wg, 100.0 * buf[k] / wg / tot))
print("total %.0f" % tot)
for k in range(9, 15):
    if buf[k]: print("stamp %d  %9.0f ticks/workgroup" % (k, buf[k] / wg))
