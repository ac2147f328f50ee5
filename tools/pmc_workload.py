#!/usr/bin/env python3
"""tools/pmc_workload.py --model contrafold|vienna --n N --batch B [--hp duplex|cofold] [--computes K]
The fixed workload of the counter passes (tools/profile_gpu.sh): one context, the batch bench.py uses for (model, n, B), one upload,
then K+1 rh_batch_compute calls and nothing else -- so that (dispatches of a kernel) / (K+1) is its launch count per step and the
counter totals divide cleanly.  Prints {"computes": K+1, ...}."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="contrafold")
ap.add_argument("--n", type=int, default=500)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--hp", default=None)
ap.add_argument("--computes", type=int, default=2)
ap.add_argument("--workload", default="pairs")
a = ap.parse_args()
import ractip_amd
from ractip_amd.seqgen import random_pairs
vienna = a.model == "vienna"
if a.workload == "zscore":
    from ractip_amd import shard
    fa = [l.strip() for l in open(os.path.join(ROOT, "ractip_amd", "data", "config5_OxyS_fhlA.fa")) if not l.startswith(">")]
    pairs = shard.zscore_shuffles(fa[0], fa[1], 12, a.batch, 1)
else:
    pairs = random_pairs(a.batch, a.n, seed=12345)
c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL if vienna else ractip_amd.hot.RH_MODEL_CONTRAFOLD)
if vienna:
    c.set_hybrid((a.hp or "cofold") == "cofold")
c.batch_upload(pairs)
for _ in range(a.computes + 1):
    c.batch_compute()
print(json.dumps({"computes": a.computes + 1, "pairs": len(pairs), "model": a.model, "n": a.n, "path": c.last_path()}))
c.close()
