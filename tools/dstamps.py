#!/usr/bin/env python3
"""tools/dstamps.py [n [pairs]]: per-phase s_memtime totals of dxl_strip8 (tuning build `python tools/build_variant.py dstamps -DRH_DSTAMPS`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACTIP_HOT_LIB"] = os.environ.get("RH_STAMPS_LIB") or os.path.join(ROOT, "ractip_amd", "libractip_hot_dstamps.so")
import numpy as np
import ractip_amd
lib = ctypes.CDLL(os.environ["RACTIP_HOT_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rng = np.random.default_rng(1)
pairs = [("".join("ACGU"[k] for k in rng.integers(0, 4, n)), "".join("ACGU"[k] for k in rng.integers(0, 4, n))) for _ in range(npairs)]
ctx = ractip_amd.Context(device=0)
ctx.batch_upload(pairs)
ctx.batch_compute()
buf = (ctypes.c_ulonglong * 16)()
lib.rh_debug_dstamps(buf, 1)
ctx.batch_compute()
lib.rh_debug_dstamps(buf, 1)
wg = max(1, buf[15])
names = ["loads issued", "window rows arrived", "staged in LDS", "cell operands + weights", "window pass", "barrier", "partial sums exchanged", "chain", "stores issued"]
tot = sum(buf[k] for k in range(9)) / wg
print("workgroups with cells:", wg)
for k in range(9):
    print("stamp %d  %-26s %8.0f ticks/workgroup  %5.1f %%" % (k, names[k], buf[k] / wg, 100.0 * buf[k] / wg / tot))
print("total %.0f" % tot)
