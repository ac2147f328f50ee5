#!/usr/bin/env python3
"""bench.py -- sequence-pairs/s of the RactIP probability-matrix hot path on MI355X.

Metric (BASELINE.json): sequence-pairs/sec incl. bp+hp+ap DP at n=500, with the
achieved HBM GB/s against the roofline.  One "step" = one pass of the hot path
(2x McCaskill inside/outside/posterior + up + duplex fw/bk/posterior) over one
batch of synthetic pairs (mt19937(12345) stream).

`value` is the SURVEY 8(d) quantity: host strings in -> thresholded matrices out.  A step is
rh_batch_upload -> rh_batch_compute -> rh_batch_candidates_all x5 (the five threshold scans of
src/ractip.cpp:557-647, compacted on the device, results on the host); steps alternate between two
contexts on two host threads so that one context's PCIe legs overlap the other's kernels.  The
device-resident rate (inputs in HBM, nothing copied back) is reported beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--seqlen 500] [--batch B]

N > 1: one process per GPU (launched by torch.distributed.run); independent pairs
are sharded across ranks (weak scaling, the z-score shard of ractip.cpp:1638-1657)
and the per-pair scalars are gathered over RCCL, the path's only exchange step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def compulsory_bytes(pairs, vienna, cofold):
    """16 B per cell of every DP table the path keeps in HBM (one store + one load), plus the result matrices: the floor for a
    kernel that held everything else on chip."""
    tabs = 16 if vienna else 13          # diagonal-major tables per sequence on the linear path (incl. block-product sums)
    tot = 0
    for s1, s2 in pairs:
        for n in (len(s1), len(s2)):
            tot += 16 * tabs * (n * (n + 1) // 2) + 8 * (n + 1) * (n + 2) // 2 + 8 * n * (15 if vienna else 1)
        if cofold:
            N = len(s1) + len(s2)
            tot += 16 * tabs * (N * (N + 1) // 2) + 8 * (N + 1) * (N + 2) // 2
        else:
            tot += 16 * (6 if vienna else 4) * len(s1) * len(s2)
        tot += 8 * (len(s1) + 1) * (len(s2) + 1)
    return tot


def cpu_baseline_vienna(pairs, budget_s=12.0, cofold=True):
    """Vienna-BL workload: our CPU restatement (oracle/vienna_oracle.c, kind "port"; ViennaRNA itself is absent), 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import ViennaOracle
    eng = ViennaOracle()
    done, t0 = 0, time.perf_counter()
    for s1, s2 in pairs:
        eng.mccaskill(s1, max_w=15)
        eng.mccaskill(s2, max_w=15)
        (eng.cofold if cofold else eng.pf_duplex)(s1, s2)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%d pair(s) of n=%d/%d, %.1f s, oracle/vienna_oracle.c (McCaskill + accessibility w<=15 + %s), 1 thread" % (
                done, len(pairs[0][0]), len(pairs[0][1]), dt, "co_pf_fold" if cofold else "pf_duplex")}


def cpu_baseline_all_cores(n, vienna, cofold, per_worker=2):
    """every host core runs the single-threaded CPU path on its own pairs (one process per core,
    no GPU use); whole-host pairs/s with the core count stated.  Bounded: `per_worker` pairs per process."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU job owns a 16-core share of its host whatever the affinity mask says; more workers only oversubscribe
    cores = min(cores, int(os.environ.get("RACTIP_BENCH_CPU_WORKERS", "16")))
    code = (
        "import sys, os, time; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from ractip_amd.seqgen import random_pairs\n"
        "k, cnt, n, vienna, cofold = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == '1', sys.argv[5] == '1'\n"
        "pairs = random_pairs((k + 1) * cnt, n, seed=12345)[k * cnt:]\n"
        "if vienna:\n"
        "    from _oracle import ViennaOracle; e = ViennaOracle()\n"
        "    for a, b in pairs: e.mccaskill(a, max_w=15); e.mccaskill(b, max_w=15); (e.cofold if cofold else e.pf_duplex)(a, b)\n"
        "else:\n"
        "    from _oracle import Oracle, Reference\n"
        "    try: e = Reference()\n"
        "    except (FileNotFoundError, OSError): e = Oracle()\n"
        "    for a, b in pairs: e.inference(a); e.inference(b); e.duplex(a, b)\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(k), str(per_worker), str(n), "1" if vienna else "0", "1" if cofold else "0"],
                              env=dict(os.environ, OMP_NUM_THREADS="1")) for k in range(cores)]
    ok = all(p.wait() == 0 for p in procs)
    dt = time.perf_counter() - t0
    if not ok:
        return None
    return {"value": cores * per_worker / dt, "unit": "pairs/s", "cores": cores,
            "sample": "%d processes x %d pair(s) of n=%d, %.1f s wall (incl. process start)" % (cores, per_worker, n, dt)}


def copy_bandwidth_GBs(torch, nbytes=1 << 30, reps=5):
    """Device-to-device copy rate (read + write bytes per second) of this GPU: the achievable-HBM figure BASELINE.md asks to
    state beside the nominal 8 TB/s."""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) / 1e3) / 1e9


def cpu_baseline(pairs, budget_s=20.0):
    """Time the CPU path on a bounded sample of the same workload (rank 0, N=1 only).

    kind "reference": the reference's own InferenceEngine<double>/DuplexEngine<double>
    (oracle/_ref, built from /root/reference in the build container); kind "port": our
    CPU restatement (oracle/libcf_oracle.so).  Single-threaded, like the reference
    (src/ractip.cpp:1494)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import Oracle, Reference
    try:
        eng, kind = Reference(), "reference"
    except (FileNotFoundError, OSError):
        eng, kind = Oracle(), "port"
    done, t0 = 0, time.perf_counter()
    for s1, s2 in pairs:
        eng.inference(s1)
        eng.inference(s2)
        eng.duplex(s1, s2)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "pairs/s", "cores": 1, "kind": kind,
            "sample": "%d pair(s) of n=%d/%d, %.1f s, InferenceEngine/DuplexEngine double, 1 thread" % (
                done, len(pairs[0][0]), len(pairs[0][1]), dt)}


SCANS = ((0, 0.5), (1, 0.5), (2, 0.1), (3, 0.003), (4, 0.003))   # which, threshold: src/cmdline.c defaults (bp1, bp2, hp, up1, up2)
FP64_PEAK_TFLOPS = 78.6   # MI355X vector / matrix FP64 (spec)


def source_hash():
    """sha256 over the kernel sources: stamps profiles/pmc_traffic.json entries, so that counter traffic measured on other
    kernels is never reported for these."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ractip_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def sweep_of(kernel):
    """Which sweep of a step a kernel name (as a kernel trace prints it) belongs to."""
    k = kernel
    if "pack_tiles<0>" in k:
        return "inside"
    if "pack_tiles<1>" in k:
        return "outside"
    if any(t in k for t in ("dxl_", "dxvl_", "dxv_", "dx_sweep", "dx_logz", "dx_posterior")):
        return "duplex"
    if any(t in k for t in ("inside", "lin_init", "f5i", "co_seed")):
        return "inside"
    if any(t in k for t in ("outside", "lin_finish", "vlin_finish", "f5o", "mc_unpaired", "_acc_")):
        return "outside"
    return "other"


def load_traffic(model, n, batch, hp=None):
    """Counter traffic of (model, n, pairs per step) from profiles/pmc_traffic.json -- only if it was measured on the current kernel
    sources.  Returns (kernels dict or None, note)."""
    key = "%s_n%d_b%d" % (model, n, batch) + ("_duplex" if hp == "duplex" else "")
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json unreadable"
    if not tj:
        return None, "not profiled (no entry %s in profiles/pmc_traffic.json)" % key
    if tj.get("source_hash") != source_hash():
        return None, "profiles/pmc_traffic.json: entry %s was measured on other kernel sources" % key
    return tj.get("kernels") or None, None


def measure_roofline(ctx, pairs, model, n, batch, vienna, cofold, hp, resident_ms, torch, with_copy):
    """The roofline object of one workload, from one context that has `pairs` uploaded and computed.
    * isolated sweeps: rh_set_overlap(0), HIP events around each sweep on its own stream (3 passes);
    * the dominant sweep's KERNEL: average launch duration measured live by event pairs around every launch of that kernel class
      (rh_set_kernel_timing; 2 passes) -- what a kernel trace of `bench.py --isolate` reports for it;
    * traffic: calibrated counter bytes (profiles/pmc_traffic.json, stamped with the kernel-source hash) of that kernel per launch, of
      EVERY kernel of the sweep, and of every kernel of the step; null when these sources were not profiled -- then `frac` is null too:
      the algorithmic figure (operand touches of the reference recurrences) exceeds the peak as soon as operands are reused on chip
      and is reported under alg_* only;
    * fp64: FLOP = algorithmic bytes / 8 (one FMA per two operand touches), and the block products' share measured on their own."""
    from ractip_amd import balg
    ctx.set_overlap(False)
    iso = np.zeros(4)
    nl = [0, 0, 0]
    for _ in range(3):
        ctx.batch_compute()
        ms, nl = ctx.batch_timings()
        iso += np.array(ms) / 3
    ctx.set_overlap(True)
    names = ctx.batch_kernels()
    # algorithmic bytes (counted exactly on the first pairs and scaled: the pairs are i.i.d. sequences of one length)
    b = {"mc_inside": 0.0, "mc_outside": 0.0, "duplex": 0.0, "total": 0.0}
    bk = {"inside": 0.0, "inside_far": 0.0, "outside": 0.0, "outside_far": 0.0}
    counted = pairs if len(pairs[0][0]) < 200 else pairs[:max(1, min(len(pairs), 8 if n <= 600 else 2))]
    scale = len(pairs) / len(counted)
    for s1, s2 in counted:
        pb = balg.pair_bytes(s1, s2)
        for k in b:
            b[k] += pb[k] * scale
        pk = balg.pair_bytes_by_kernel(s1, s2)
        for k in bk:
            bk[k] += pk[k] * scale
    if cofold:
        sample = pairs[:2] if n <= 600 else pairs[:1]
        co = sum(8 * sum(sum(v) for v in balg.mccaskill_counts(s1 + s2).values()) for s1, s2 in sample) * len(pairs) / len(sample)
        b["total"] += co - b["duplex"]
        b["duplex"] = co
    sweeps = {}
    for idx, (sname, bytes_) in enumerate((("inside", b["mc_inside"]), ("outside", b["mc_outside"]), ("duplex", b["duplex"]))):
        fine, far, n_far = names[idx]
        if not fine or nl[idx] == 0 or iso[idx] <= 0:
            continue
        sweeps[sname] = {"kernel": fine, "block_product_kernel": far or None, "launches_per_step": int(nl[idx]), "of_which_block_product": int(n_far),
                         "isolated_ms_per_step": float(iso[idx]), "alg_GB_per_step": bytes_ / 1e9,
                         "fp64_TFLOPs": bytes_ / 8.0 / (iso[idx] / 1e3) / 1e12}
    dom = max(sweeps, key=lambda k: sweeps[k]["isolated_ms_per_step"])
    sd = sweeps[dom]
    cls = {"inside": 0, "outside": 2, "duplex": 4}[dom]
    k_us, k_launches = ctx.kernel_class_times(cls, computes=2)
    if not k_launches:   # (the two-molecule sweeps of the Vienna-BL model run under the McCaskill classes: the sweep's own event timing instead)
        k_us, k_launches = sd["isolated_ms_per_step"] * 1e3 / sd["launches_per_step"], float(sd["launches_per_step"])
    far_us = far_launches = None
    if sd["of_which_block_product"] and dom != "duplex":   # (the two-molecule sweeps of the Vienna-BL model share classes 0-3 with the folds)
        far_us, far_launches = ctx.kernel_class_times(cls + 1, computes=2)
    alg_fine = bk.get(dom, sd["alg_GB_per_step"] * 1e9) if dom != "duplex" else b["duplex"]
    kern, note = load_traffic(model, n, batch, hp)
    per = lambda name: next((v for k2, v in (kern or {}).items() if name and name.split("(")[0] in k2), None)
    kt = per(sd["kernel"])
    traffic = kt["bytes_per_dispatch"] if kt else None
    sweep_bytes = step_bytes = None
    by_kernel = {}
    if kern:
        step_bytes = sum(v["bytes_per_dispatch"] * v["dispatches_per_step"] for v in kern.values())
        sweep_bytes = sum(v["bytes_per_dispatch"] * v["dispatches_per_step"] for k2, v in kern.items() if sweep_of(k2) == dom)
        by_kernel = {k2.replace("void ", "").replace("rh::", ""): {"sweep": sweep_of(k2), "MB_per_dispatch": round(v["bytes_per_dispatch"] / 1e6, 3),
                                                                      "dispatches_per_step": round(v["dispatches_per_step"], 2)} for k2, v in kern.items()}
    launch_s = k_us * 1e-6
    ach = traffic / 1e9 / launch_s if traffic else None
    comp_step = compulsory_bytes(pairs, vienna, cofold)
    fp64_whole = b["total"] / 8.0 / (resident_ms / 1e3) / 1e12
    hbm_whole = (step_bytes / 1e9 / (resident_ms / 1e3)) if step_bytes else None
    far_flops = None
    if far_us:
        # block products of the dominant sweep: FLOP = 2 x (k-terms the product kernels take over), time = their launches alone (live events)
        far_bytes = bk.get(dom + "_far", 0.0)
        far_flops = {"GFLOP_per_step": far_bytes / 8.0 / 1e9, "ms_per_step": far_us * far_launches / 1e3, "launches_per_step": far_launches,
                     "TFLOPs": far_bytes / 8.0 / (far_us * far_launches * 1e-6) / 1e12}
        far_flops["fp64_frac"] = far_flops["TFLOPs"] / FP64_PEAK_TFLOPS
    hbm_frac = ach / HBM_PEAK_GBS if ach else None
    fp_frac_k = alg_fine / 8.0 / (k_us * 1e-6 * k_launches) / 1e12 / FP64_PEAK_TFLOPS if k_launches else None
    bound = "hbm"
    if hbm_whole is not None and fp64_whole / FP64_PEAK_TFLOPS > hbm_whole / HBM_PEAK_GBS:
        bound = "fp64"
    r = {"bound": bound, "kernel": sd["kernel"], "sweep": dom,
         "avg_launch_us": k_us, "launches_per_step": k_launches,
         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac, "traffic": traffic, "traffic_note": note,
         "achieved_basis": "calibrated rocprofv3 counter bytes of this kernel per launch (profiles/pmc_traffic.json) / its live event-pair launch duration" if traffic
                           else "no counter profile of these kernel sources: achieved and frac are null (the algorithmic figure is under alg_*)",
         "alg_bytes_per_launch": alg_fine / k_launches if k_launches else None,
         "alg_GBs": alg_fine / 1e9 / (k_us * 1e-6 * k_launches) if k_launches else None,
         "alg_over_peak": alg_fine / 1e9 / (k_us * 1e-6 * k_launches) / HBM_PEAK_GBS if k_launches else None,
         "fp64_frac": fp_frac_k,
         "sweep_level": {"isolated_ms_per_step": sd["isolated_ms_per_step"], "launches_per_step": sd["launches_per_step"],
                         "traffic_GB_per_step": sweep_bytes / 1e9 if sweep_bytes else None,
                         "achieved_GBs": sweep_bytes / 1e9 / (sd["isolated_ms_per_step"] / 1e3) if sweep_bytes else None,
                         "frac": sweep_bytes / 1e9 / (sd["isolated_ms_per_step"] / 1e3) / HBM_PEAK_GBS if sweep_bytes else None,
                         "block_products": far_flops},
         "whole_path": {"ms_per_step_device_resident": resident_ms, "traffic_GB_per_step": step_bytes / 1e9 if step_bytes else None,
                        "achieved_GBs": hbm_whole, "hbm_frac": hbm_whole / HBM_PEAK_GBS if hbm_whole else None,
                        "compulsory_GB_per_step": comp_step / 1e9, "traffic_over_compulsory": step_bytes / comp_step if step_bytes else None,
                        "alg_GB_per_pair": b["total"] / 1e9 / max(1, len(pairs)), "fp64_TFLOPs": fp64_whole, "fp64_frac": fp64_whole / FP64_PEAK_TFLOPS},
         "peak_fp64_TFLOPs": FP64_PEAK_TFLOPS,
         "peak_copy_measured": copy_bandwidth_GBs(torch) if with_copy else None,
         "timing": "avg_launch_us: HIP event pairs around every launch of the kernel, sweeps not overlapping (2 passes after the timed region) = "
                   "the per-kernel average of `rocprofv3 --kernel-trace --stats -- python3 bench.py --isolate` (profiles/*_kernel_stats.txt); "
                   "sweeps: HIP events around each whole sweep, 3 passes",
         "sweeps": sweeps, "traffic_by_kernel": by_kernel}
    return r


def timed_steps(ctxs, pairs, steps, body, fence):
    """`steps` steps, step k on context k % len(ctxs), one host thread per context; wall time of all of them."""
    import threading
    err = []

    def work(k0):
        try:
            for k in range(k0, steps, len(ctxs)):
                body(ctxs[k0], pairs)
        except Exception as e:   # noqa: BLE001 -- re-raised below
            err.append(e)
    fence()
    t0 = time.perf_counter()
    if len(ctxs) == 1:
        work(0)
    else:
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(ctxs))]
        [t.start() for t in th]
        [t.join() for t in th]
    fence()
    dt = time.perf_counter() - t0
    if err:
        raise err[0]
    return dt


def step_candidates(ctx, pairs):
    ctx.batch_upload(pairs)
    ctx.batch_compute()
    for which, th in SCANS:
        ctx.batch_candidates_all(which, th)


def quick_rates(ractip_amd, torch, device_index, pairs, steps, model="contrafold", hp=None):
    """device-resident and end-to-end (upload -> compute -> 5 scans on the host, two contexts) rates of one more workload, with its
    own roofline object"""
    vienna = model == "vienna"
    cofold = vienna and (hp or "cofold") == "cofold"
    mid = ractip_amd.hot.RH_MODEL_VIENNA_BL if vienna else ractip_amd.hot.RH_MODEL_CONTRAFOLD
    ctxs = [ractip_amd.Context(device=device_index, model=mid) for _ in range(2)]
    try:
        for c in ctxs:
            if vienna:
                c.set_hybrid(cofold)
            c.batch_upload(pairs)
            c.batch_compute()
        fence = torch.cuda.synchronize
        dt_res = timed_steps(ctxs[:1], pairs, steps, lambda c, p: c.batch_compute(), fence)
        ms, nl = ctxs[0].batch_timings()
        dt_e2e = timed_steps(ctxs, pairs, 2 * steps, step_candidates, fence)
        n = max(len(pairs[0][0]), len(pairs[0][1]))
        out = {"pairs_per_step": len(pairs), "steps": steps, "model": model,
               "device_resident_pairs_per_s": len(pairs) * steps / dt_res, "ms_per_step_device_resident": dt_res / steps * 1e3,
               "value_pairs_per_s": len(pairs) * 2 * steps / dt_e2e,
               "phase_ms": {"mccaskill_inside": ms[0], "mccaskill_outside": ms[1], "duplex": ms[2]},
               "path": ctxs[0].last_path()}
        try:
            out["roofline"] = measure_roofline(ctxs[0], pairs, model, n, len(pairs), vienna, cofold, hp, dt_res / steps * 1e3, torch, False)
        except Exception as e:   # noqa: BLE001 -- the rates above stay
            out["roofline_error"] = repr(e)
        return out
    finally:
        for c in ctxs:
            c.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seqlen", "--n", dest="n", type=int, default=500,
                    help="sequence length of the synthetic pairs (BASELINE config 3: 500, config 4: 2000)")
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = auto)")
    ap.add_argument("--contexts", type=int, default=0, help="contexts / host threads whose steps alternate on one GPU (0 = auto: 2 for n <= 1200 on one GPU, else 1)")
    ap.add_argument("--workload", default="pairs", choices=["pairs", "zscore"],
                    help="pairs: synthetic random pairs of one length; zscore: BASELINE config 5, the DP stage of the z-score loop "
                         "(src/ractip.cpp:1638-1657): 1000 dinucleotide shuffles of OxyS/fhlA per step")
    ap.add_argument("--total", type=int, default=0,
                    help="zscore workload: FIXED total number of shuffles split over the ranks (strong scaling, BASELINE config 5 "
                         "as stated: 1000); 0 = --batch shuffles per rank (weak scaling)")
    ap.add_argument("--model", default="contrafold", choices=["contrafold", "vienna"],
                    help="contrafold: the in-tree engines (parity pinned); vienna: the default CLI path (Vienna-BL, parity unpinned)")
    ap.add_argument("--hp", default=None, choices=["duplex", "cofold"],
                    help="vienna only: hp from pf_duplex (the --duplex branch) or from the two-molecule ensemble (default branch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra n=2000 / z-score / Vienna-BL measurements of the default run")
    ap.add_argument("--isolate", action="store_true", help="profiling runs: the sweeps of every step run one after the other (no stream overlap), so a "
                    "kernel trace of this command reports the isolated per-kernel durations that roofline.avg_launch_us measures live")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for "
                    "rehearsing the multi-rank path on a one-GPU box, where all ranks share cuda:0)")
    args = ap.parse_args()

    import torch
    import ractip_amd
    from ractip_amd import balg
    from ractip_amd.seqgen import random_pairs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback to measure)")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(args.backend)

    n = args.n
    scaling = "weak"
    if args.workload == "zscore":
        # bundled sequences of the reference (data/OxyS.fa, data/fhlA.fa), shuffled exactly as ractip.cpp:1636-1643 does
        from ractip_amd import shard as _shard
        fa = [l.strip() for l in open(os.path.join(ROOT, "ractip_amd", "data", "config5_OxyS_fhlA.fa")) if not l.startswith(">")]
        oxys, fhla = fa[0], fa[1]
        n = max(len(oxys), len(fhla))
        if args.total:   # BASELINE config 5 as stated: a fixed number of shuffles over the ranks, contiguous iteration blocks
            scaling = "strong"
            all_pairs = _shard.zscore_shuffles(oxys, fhla, 12, args.total, 1)
            lo, hi = _shard.shard_bounds(args.total, rank, world)
            pairs = all_pairs[lo:hi]
            batch = hi - lo
        else:
            batch = args.batch or 1000
            all_pairs = _shard.zscore_shuffles(oxys, fhla, 12, batch * world, 1)
            pairs = all_pairs[rank * batch:(rank + 1) * batch]
    else:
        # measured on MI355X (r02, CONTRAfold model): n=500: 9029 / 9337 / 9506 / 9571 pairs/s at 256 / 384 / 512 / 768 pairs per step (a strip
        # launch is 8 rounds of workgroups at 256 pairs: the tail of every launch weighs less on a longer one); n=2000: 308 at 32, 490 at 64
        batch = args.batch or (512 if n <= 600 else (64 if n <= 1200 else 32))
        if args.model == "vienna" and not args.batch:
            # n=500: 1763 / 1816 / 1835 pairs/s at 128 / 192 / 256 (2 contexts x 61 GB of tables at 256); n=2000: 48 pairs/s at 32, 42 at 16 (125 GB)
            batch = 256 if n <= 600 else (max(1, batch // 2) if n <= 1200 else batch)
        # every rank draws from ONE stream and keeps its own slice: distinct pairs per rank (weak scaling)
        all_pairs = random_pairs(batch * world, n, seed=12345)
        pairs = all_pairs[rank * batch:(rank + 1) * batch]
    total_per_step = len(all_pairs)

    vienna = args.model == "vienna"
    model_id = ractip_amd.hot.RH_MODEL_VIENNA_BL if vienna else ractip_amd.hot.RH_MODEL_CONTRAFOLD
    cofold = vienna and (args.hp or "cofold") == "cofold"
    # two contexts on one GPU: their steps alternate, so uploads / result copies of one overlap the kernels of the other.
    # Ranks of a multi-GPU run keep one context: the gather is a collective and stays on the main thread.
    n_ctx = args.contexts or (2 if (world == 1 and n <= 1200 and not args.isolate) else 1)   # (--isolate: nothing may overlap a timed kernel)
    ctxs = [ractip_amd.Context(device=device_index, model=model_id) for _ in range(n_ctx)]
    for c in ctxs:
        if vienna:
            c.set_hybrid(cofold)
        c.batch_upload(pairs)   # tables allocated, launch graphs captured: outside the timed region
        c.batch_compute()
        if args.isolate:
            c.set_overlap(False)
    ctx = ctxs[0]

    from ractip_amd import shard

    def step(c, p):
        step_candidates(c, p)   # host strings -> HBM, all DP kernels, the five threshold scans back on the host
        if dist is not None:    # the shard's only exchange: per-pair scalars to every rank (ractip.cpp:1655-1663)
            shard.gather_in_order(c.batch_logz(), total_per_step, dist,
                                  device=torch.device("cuda", device_index) if args.backend == "nccl" else None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(ctxs[k % n_ctx], pairs)
    dt = timed_steps(ctxs, pairs, args.steps, step, fence)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # device-resident rate (inputs in HBM, nothing copied back): one context, the same number of steps
    ms_acc = np.zeros(4)
    nl = [0, 0, 0]

    def resident(c, p):
        nonlocal nl
        c.batch_compute()
        ms, nl = c.batch_timings()
        ms_acc[:] += np.array(ms)
    dt_res = timed_steps(ctxs[:1], pairs, args.steps, resident, fence)

    if rank == 0:
        total_pairs = total_per_step * args.steps
        roof = measure_roofline(ctx, pairs, args.model, n, batch, vienna, cofold, args.hp, dt_res / args.steps * 1e3, torch, world == 1)
        if args.isolate:
            ctx.set_overlap(False)
        ms_mean = ms_acc / args.steps  # HIP-event ms per step inside the device-resident loop: inside sweep, outside sweep, duplex, whole
        roof["phases_in_loop"] = {"mccaskill_inside_ms": float(ms_mean[0]), "mccaskill_outside_ms": float(ms_mean[1]), "duplex_ms": float(ms_mean[2]),
                                  "launches": [int(x) for x in nl], "note": "the duplex stream overlaps the McCaskill stream here"}
        line = {
            "metric": ("sequence-pairs/sec (incl. bp+hp+ap DP; host strings in, thresholded matrices out) at n=%d" % n) if args.workload == "pairs"
                      else "z-score DP stage: shuffled pairs/sec (OxyS/fhlA, bp+hp+ap DP per shuffle; host strings in, thresholded matrices out)",
            "value": total_pairs / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "device_resident_pairs_per_s": len(pairs) * args.steps / dt_res,
            "device_resident_ms_per_step": dt_res / args.steps * 1e3,
            "config": {"workload": ("synthetic random pairs n=%d/%d, std::mt19937(12345) stream (BASELINE config %s)"
                                    % (n, n, "3" if n == 500 else ("4" if n == 2000 else "-"))) if args.workload == "pairs"
                                   else "OxyS.fa (109) vs fhlA.fa (113), --zscore=12 --seed=1 dinucleotide shuffles (BASELINE config 5, DP stage)",
                       "pairs_per_gpu_per_step": batch, "total_pairs_per_step": total_per_step, "model": None,
                       "step": "rh_batch_upload -> rh_batch_compute -> rh_batch_candidates_all x5 (thresholds 0.5/0.5/0.1/0.003/0.003)"
                               + (", steps alternate between %d contexts / host threads" % n_ctx if n_ctx > 1 else "")
                               + (", + all_gather of 3 doubles per pair" if world > 1 else ""),
                       "scoring": ("Vienna-BL (BL* tables, ViennaRNA-1.8 semantics, up width 15, hp from %s; parity unpinned)"
                                   % ("co_pf_fold(s1+s2)" if cofold else "pf_duplex")) if vienna
                                  else "CONTRAfold complementary (708 weights)"},
            "roofline": roof,
        }
        if world == 1:
            # dense results instead of the thresholded lists: all five matrices into page-locked host buffers, two contexts
            bufs = {}

            def step_dense(c, p):
                c.batch_upload(p)
                c.batch_compute()
                bufs[id(c)] = c.batch_results_all_into(bufs.get(id(c)))
            for c in ctxs:
                step_dense(c, pairs)
            k_dense = max(2, min(args.steps, 6))
            dt_dense = timed_steps(ctxs, pairs, k_dense, step_dense, fence)
            line["pcie_inclusive"] = {"threshold_candidates_pairs_per_s": line["value"],
                                      "dense_results_pairs_per_s": len(pairs) * k_dense / dt_dense,
                                      "dense_GB_per_step": sum(a.nbytes for a in bufs[id(ctx)]) / 1e9}
        for c in ctxs:
            c.close()
        ctxs = []
        if world == 1 and not args.no_also and args.workload == "pairs" and n == 500 and not vienna:
            # the other BASELINE configurations, bounded to a few seconds each: config 4 (n = 2000) and config 5 (z-score DP stage)
            also = {}
            try:
                also["n2000"] = quick_rates(ractip_amd, torch, device_index, random_pairs(64, 2000, seed=12345), 3)   # 64 pairs: ~60 GB of tables per context
                from ractip_amd import shard as _shard
                fa = [l.strip() for l in open(os.path.join(ROOT, "ractip_amd", "data", "config5_OxyS_fhlA.fa")) if not l.startswith(">")]
                also["zscore_1000_shuffles"] = quick_rates(ractip_amd, torch, device_index, _shard.zscore_shuffles(fa[0], fa[1], 12, 1000, 1), 10)
                # the reference's DEFAULT path (Vienna-BL model, parity unpinned): rnafold x2 + accessibility + hp from co_pf_fold / pf_duplex
                also["vienna_n500_cofold"] = quick_rates(ractip_amd, torch, device_index, random_pairs(256, 500, seed=12345), 3, model="vienna", hp="cofold")
                also["vienna_n500_duplex"] = quick_rates(ractip_amd, torch, device_index, random_pairs(256, 500, seed=12345), 3, model="vienna", hp="duplex")
            except Exception as e:   # noqa: BLE001 -- the headline line must still be printed
                also["error"] = repr(e)
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            if vienna:
                line["cpu_baseline"] = cpu_baseline_vienna(all_pairs, budget_s=12.0 if n <= 600 else 1.0, cofold=cofold)
            else:
                line["cpu_baseline"] = cpu_baseline(all_pairs, budget_s=12.0 if n <= 600 else 1.0)
            if n <= 600 and args.workload == "pairs":
                line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(n, vienna, cofold, per_worker=1 if vienna and cofold else 2)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
