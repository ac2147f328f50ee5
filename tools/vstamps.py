#!/usr/bin/env python3
"""tools/vstamps.py [pairs]: per-phase s_memtime totals of the Vienna-BL look-ahead kernel vlin_inside_diag<8, 16, CUT, 1> (library built
by `python tools/build_variant.py vstamps -DRH_VSTAMPS=1` for the single-molecule kernel, `=2` for the two-molecule one); prints the
average ticks per cell workgroup between consecutive RH_VSTAMP sites (mccaskill_vlin.hip), wavefront 0 of each workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACTIP_HOT_LIB"] = os.environ.get("RH_STAMPS_LIB") or os.path.join(ROOT, "ractip_amd", "libractip_hot_vstamps.so")
import ractip_amd
from ractip_amd.seqgen import random_pairs
lib = ctypes.CDLL(os.environ["RACTIP_HOT_LIB"])
pairs = random_pairs(int(sys.argv[1]) if len(sys.argv) > 1 else 128, 500, seed=12345)
c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
c.set_hybrid(True)
c.set_overlap(False)
c.batch_upload(pairs)
c.batch_compute()
buf = (ctypes.c_ulonglong * 16)()
lib.rh_debug_vstamps(buf, 1)
c.batch_compute()
lib.rh_debug_vstamps(buf, 1)
wg = max(1, buf[15])
names = ["letters + pair type", "FM2 near terms", "staging -> LDS", "filters", "barrier + exchange", "epilogue operands", "epilogue + stores"]
tot = sum(buf[k] for k in range(7)) / wg
print("cell workgroups:", wg)
for k in range(7):
    print("%-22s %9.0f ticks/workgroup  %5.1f %%" % (names[k], buf[k] / wg, 100.0 * buf[k] / wg / max(tot, 1)))
print("%-22s %9.0f" % ("total (wavefront 0)", tot))
c.close()
