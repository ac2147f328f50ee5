#!/usr/bin/env python3
"""tools/trace_strips.py [n [pairs]]: per-workgroup timeline of one inside strip launch (tuning build `python tools/build_variant.py
trace -DRH_STAMPS=3`): are the two workgroups of a CU in lock-step, how long does staging take against how many workgroups stage
at the same moment, how much of the launch does a CU spend with nobody computing."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RACTIP_HOT_LIB"] = os.environ.get("RH_STAMPS_LIB") or os.path.join(ROOT, "ractip_amd", "libractip_hot_trace.so")
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
ctx.batch_compute()
buf = (ctypes.c_ulonglong * (5 * 16384))()
assert lib.rh_debug_trace(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 5)
a = a[a[:, 1] > 0]
ident = a[:, 0]
hw = (ident & np.uint64(0xffffffff)).astype(np.int64)
xcc = (ident >> np.uint64(32)).astype(np.int64) & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)   # cu_id, sh_id, se_id, xcc
t = a[:, 1:].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) * 24.0   # 100 MHz ticks -> ~shader cycles at 2.4 GHz
print("workgroups traced:", len(t), " distinct CUs:", len(set(cu.tolist())), " launch span %.0f kcycles" % (t[:, 3].max() / 1e3))
stage, pre, chain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print("stage %.1f k (p10 %.1f, p90 %.1f)   pre-phase %.1f k   chain+stores %.1f k   life %.1f k" % (
    stage.mean() / 1e3, np.percentile(stage, 10) / 1e3, np.percentile(stage, 90) / 1e3, pre.mean() / 1e3, chain.mean() / 1e3, (t[:, 3] - t[:, 0]).mean() / 1e3))
# how many workgroups of the whole chip are staging when a workgroup starts staging
starts, ends = np.sort(t[:, 0]), np.sort(t[:, 1])
conc = np.searchsorted(starts, t[:, 0], side="right") - np.searchsorted(ends, t[:, 0], side="right")
for lo, hi in ((0, 64), (64, 128), (128, 256), (256, 384), (384, 10000)):
    m = (conc >= lo) & (conc < hi)
    if m.any():
        print("  %4d..%-5d staging chip-wide at its start: %5d workgroups, stage %.1f k" % (lo, hi, m.sum(), stage[m].mean() / 1e3))
# per CU: fraction of time with 0 / 1 / 2 workgroups in their compute phases, and partner phase offset
idle = []
off = []
for c in set(cu.tolist()):
    m = np.where(cu == c)[0]
    if len(m) < 4:
        continue
    ev = []
    for k in m:
        ev.append((t[k, 1], 1)); ev.append((t[k, 3], -1))
    ev.sort()
    lo, hi = t[m, 0].min(), t[m, 3].max()
    cur, last, busy0 = 0, lo, 0.0
    for tt, d in ev:
        if cur == 0:
            busy0 += tt - last
        cur += d
        last = tt
    idle.append(busy0 / (hi - lo))
    s = np.sort(t[m, 0])
    life = (t[m, 3] - t[m, 0]).mean()
    d = np.diff(s)
    off.append(np.median(d) / life)
print("per CU: fraction of the launch with NO workgroup in a compute phase: mean %.2f (p10 %.2f p90 %.2f)" % (np.mean(idle), np.percentile(idle, 10), np.percentile(idle, 90)))
print("per CU: median gap between consecutive workgroup starts / workgroup life: %.2f (0 = lock-step pairs, 0.5 = anti-phase)" % np.mean(off))
