#!/usr/bin/env python3
"""Timing helper: HIP-event phase times of rh_batch_compute for synthetic pairs (optionally with an alternative
build of the library; RH_EXP_SKIP=dx|mc times one engine alone).  usage: exp_time.py default|lib.so n batch [mode]"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ractip_amd.hot as hot
from ractip_amd.seqgen import random_pairs
lib, n, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if lib != "default":
    import shutil
    dst = os.path.join(ROOT, "ractip_amd", "data", "..", "libractip_hot_exp.so")
    hot.LIB_PATH = os.path.abspath(lib)
    # the library locates its weights next to itself
    os.makedirs(os.path.join(os.path.dirname(hot.LIB_PATH), "data"), exist_ok=True)
    shutil.copy(os.path.join(ROOT, "ractip_amd", "data", "contrafold_complementary.params"),
                os.path.join(os.path.dirname(hot.LIB_PATH), "data"))
ctx = hot.Context()
ctx.set_mode(mode)
ctx.batch_upload(random_pairs(batch, n))
import numpy as np
acc = []
for it in range(6):
    ctx.batch_compute()
    ms, nl = ctx.batch_timings()
    if it >= 2: acc.append(ms)
m = np.mean(acc, axis=0)
print("%-28s n=%d batch=%d mode=%d inside %.2f outside %.2f duplex %.2f whole %.2f ms (path %d)" % (
    os.path.basename(lib), n, batch, mode, m[0], m[1], m[2], m[3], ctx.last_path()), flush=True)
