#!/usr/bin/env python3
"""bench.py -- sequence-pairs/s of the RactIP probability-matrix hot path on MI355X.

Metric (BASELINE.json): sequence-pairs/sec incl. bp+hp+ap DP at n=500, with the
achieved algorithmic HBM GB/s against the roofline.  One "step" = one pass of the
hot path (2x McCaskill inside/outside/posterior + up + duplex fw/bk/posterior)
over one batch of synthetic pairs (mt19937(12345) stream),
inputs already resident in HBM when the timed region starts.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seqlen", "--n", dest="n", type=int, default=500,
                    help="sequence length (BASELINE config 3: 500; config 4: 2000); use --seqlen under torch.distributed.run")
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = auto)")
    ap.add_argument("--workload", default="pairs", choices=["pairs", "zscore"],
                    help="pairs: synthetic random pairs of --seqlen (the headline metric); zscore: the DP stage of BASELINE "
                         "config 5 -- OxyS vs fhlA, --zscore=12 --seed=1, --batch dinucleotide shuffles per GPU per step")
    ap.add_argument("--model", default="contrafold", choices=["contrafold", "vienna"],
                    help="contrafold: the pinned --contrafold path (headline); vienna: the default-CLI path with --duplex "
                         "(pf_fold bp + pf_unstru up at width 15 + pf_duplex hp, BL* energies, parity unpinned)")
    ap.add_argument("--hp", default=None, choices=["duplex", "cofold"],
                    help="--model vienna only: hybridization matrix from pf_duplex (the --duplex branch) or from the two-molecule "
                         "ensemble co_pf_fold(s1+s2), RactIP's default (default: cofold)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if args.workload == "zscore":
        # bundled sequences of the reference (data/OxyS.fa, data/fhlA.fa), shuffled exactly as ractip.cpp:1636-1643 does
        from ractip_amd import shard as _shard
        fa = [l.strip() for l in open(os.path.join(ROOT, "ractip_amd", "data", "config5_OxyS_fhlA.fa")) if not l.startswith(">")]
        oxys, fhla = fa[0], fa[1]
        batch = args.batch or 1000
        n = max(len(oxys), len(fhla))
        all_pairs = _shard.zscore_shuffles(oxys, fhla, 12, batch * world, 1)
    else:
        # measured on MI355X (r01_n): n=500: 6218 pairs/s at 256 pairs/step, 5976 at 128, 5182 at 64, 5995 at 512;
        # n=2000: 271 at 32, 202 at 16; Vienna-BL n=500: 1007 at 128, 960 at 64
        batch = args.batch or (256 if n <= 600 else (64 if n <= 1200 else 32))
        if args.model == "vienna" and not args.batch:
            batch = max(1, batch // 2) if n <= 1200 else batch   # n=2000: 48 pairs/s at 32, 42 at 16 (125 GB of tables)
        # every rank draws from ONE stream and keeps its own slice: distinct pairs per rank (weak scaling)
        all_pairs = random_pairs(batch * world, n, seed=12345)
    pairs = all_pairs[rank * batch:(rank + 1) * batch]

    vienna = args.model == "vienna"
    ctx = ractip_amd.Context(device=device_index, model=ractip_amd.hot.RH_MODEL_VIENNA_BL if vienna else ractip_amd.hot.RH_MODEL_CONTRAFOLD)
    cofold = vienna and (args.hp or "cofold") == "cofold"
    if vienna:
        ctx.set_hybrid(cofold)
    ctx.batch_upload(pairs)  # sequences -> HBM, tables allocated: outside the timed region

    from ractip_amd import shard

    def step():
        ctx.batch_compute()  # all DP kernels, blocks until the device is done
        if dist is not None:  # the shard's only exchange: per-pair scalars to every rank (ractip.cpp:1655-1663)
            shard.gather_in_order(ctx.batch_logz(), batch * world, dist,
                                  device=torch.device("cuda", device_index) if args.backend == "nccl" else None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ms_acc = np.zeros(4)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms, nl = ctx.batch_timings()
        ms_acc += np.array(ms)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # isolated phase timings (outside the timed region): the duplex sweeps, the McCaskill inside sweep and the outside sweep
    # run one after the other, each bracketed by HIP events on its own stream, with nothing else on the device --
    # the per-kernel durations a kernel trace of this command reports (profiles/*_kernel_stats.txt)
    ctx.set_overlap(False)
    iso_ms = np.zeros(4)
    ISO = 3
    for _ in range(ISO):
        ctx.batch_compute()
        iso_ms += np.array(ctx.batch_timings()[0]) / ISO
    ctx.set_overlap(True)
    kernel_names = ctx.batch_kernels()

    if rank == 0:
        total_pairs = batch * world * args.steps
        # algorithmic bytes of rank 0's batch, split by kernel (ractip_amd/balg.py)
        # (random-pair workload: counted exactly on the first pairs and scaled -- the pairs are i.i.d. sequences of one length;
        #  the z-score workload is counted in full)
        b = {"mc_inside": 0, "mc_outside": 0, "duplex": 0, "total": 0}
        counted = pairs if args.workload == "zscore" else pairs[:max(1, min(len(pairs), 8 if n <= 600 else 2))]
        scale = len(pairs) / len(counted)
        for s1, s2 in counted:
            pb = balg.pair_bytes(s1, s2)
            for k in b:
                b[k] += pb[k] * scale
        if cofold:
            # hp comes from the McCaskill recurrences over s1+s2: count those instead of the duplex sweeps (first two
            # pairs counted exactly, scaled to the batch: the pairs are i.i.d. random sequences of one length)
            sample = pairs[:2] if n <= 600 else pairs[:1]
            co = sum(8 * sum(sum(v) for v in balg.mccaskill_counts(s1 + s2).values()) for s1, s2 in sample) * len(pairs) / len(sample)
            b["total"] += co - b["duplex"]
            b["duplex"] = co
        ms_mean = ms_acc / args.steps  # HIP-event ms per step: inside sweep, outside sweep, duplex, whole
        phases = {}
        for key, bytes_, ms_k, launches in (("mccaskill_inside_phase", b["mc_inside"], ms_mean[0], nl[0]),
                                            ("mccaskill_outside_phase", b["mc_outside"], ms_mean[1], nl[1]),
                                            ("duplex_phase", b["duplex"], ms_mean[2], nl[2])):
            phases[key] = {"alg_GB_per_step": bytes_ / 1e9, "ms_per_step": float(ms_k), "launches": int(launches),
                           "avg_launch_us": float(ms_k) * 1e3 / max(1, launches),
                           "achieved_GBs": bytes_ / 1e9 / (ms_k / 1e3) if ms_k > 0 else None}
        # roofline of the dominant sweep: the phase with the largest isolated device time.  Its launches are the
        # per-diagonal kernel plus (fast path) the block-product kernel that takes the far k-terms of the same sums;
        # algorithmic bytes per launch = B_alg of the sweep / its launches, duration = isolated sweep time / its launches
        bk = {"inside": 0, "inside_far": 0, "outside": 0, "outside_far": 0, "duplex": 0}
        far_on = kernel_names[0][2] > 0
        for s1, s2 in counted:
            pk = balg.pair_bytes_by_kernel(s1, s2, bs=16 if far_on else 0)
            for k in bk:
                bk[k] += pk[k] * scale
        if cofold:
            bk["duplex"] = b["duplex"]
        kernels = {}
        for idx, (pname, bytes_) in enumerate((("mccaskill_inside_phase", b["mc_inside"]), ("mccaskill_outside_phase", b["mc_outside"]),
                                               ("duplex_phase", b["duplex"]))):
            fine, far, n_far = kernel_names[idx]
            launches = int(nl[idx])
            if not fine or launches == 0 or iso_ms[idx] <= 0:
                continue
            avg_us = iso_ms[idx] * 1e3 / launches
            kernels[fine] = {"phase": pname, "launches_per_step": launches, "of_which_block_product": n_far,
                             "block_product_kernel": far or None, "isolated_ms_per_step": float(iso_ms[idx]),
                             "avg_launch_us": avg_us, "alg_bytes_per_launch": bytes_ / launches,
                             "achieved_GBs": bytes_ / 1e9 / (iso_ms[idx] / 1e3),
                             "alg_GB_fine_vs_block_product": [bk[("inside", "outside", "duplex")[idx]] / 1e9,
                                                              bk.get(("inside_far", "outside_far", "none")[idx], 0) / 1e9]}
        dom = max(kernels, key=lambda k: kernels[k]["isolated_ms_per_step"])
        # HBM-side traffic per launch of that sweep: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction),
        # collected separately (profiles/pmc_traffic.json, bytes per dispatch by kernel name); null if not profiled
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("%s_n%d_b%d" % (args.model, n, batch), {})
            kd = kernels[dom]
            def per_dispatch(name):
                hits = [v for k2, v in tj.items() if name and name.split("<")[0] in k2 and ("mfma" in name) == ("mfma" in k2)]
                return hits[0] if hits else None
            tf, tb = per_dispatch(dom), per_dispatch(kd["block_product_kernel"])
            if tb is not None and (kd["block_product_kernel"] or "").endswith("_pk"):
                tb += per_dispatch("lin_pack_tiles") or 0.0   # the operand-packing launch that precedes each product
            if tf is not None and (tb is not None or not kd["of_which_block_product"]):
                nfine = kd["launches_per_step"] - kd["of_which_block_product"]
                traffic = (tf * nfine + (tb or 0.0) * kd["of_which_block_product"]) / kd["launches_per_step"]
        except (OSError, ValueError):
            pass
        ach = kernels[dom]["achieved_GBs"]
        line = {
            "metric": ("sequence-pairs/sec (incl. bp+hp+ap DP) at n=%d" % n) if args.workload == "pairs"
                      else "z-score DP stage: shuffled pairs/sec (OxyS/fhlA, bp+hp+ap DP per shuffle)",
            "value": total_pairs / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("synthetic random pairs n=%d/%d, std::mt19937(12345) stream (BASELINE config %s)"
                                    % (n, n, "3" if n == 500 else ("4" if n == 2000 else "-"))) if args.workload == "pairs"
                                   else "OxyS.fa (109) vs fhlA.fa (113), --zscore=12 --seed=1 dinucleotide shuffles (BASELINE config 5, DP stage)",
                       "pairs_per_gpu_per_step": batch, "model": None,
                       "scoring": ("Vienna-BL (BL* tables, ViennaRNA-1.8 semantics, up width 15, hp from %s; parity unpinned)"
                                   % ("co_pf_fold(s1+s2)" if cofold else "pf_duplex")) if vienna
                                  else "CONTRAfold complementary (708 weights)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "peak_copy_measured": copy_bandwidth_GBs(torch) if world == 1 else None,
                         "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": traffic,
                         "alg_bytes_per_launch": kernels[dom]["alg_bytes_per_launch"],
                         "avg_launch_us": kernels[dom]["avg_launch_us"],
                         "timing": "HIP events around the whole sweep on its own stream, sweeps run one after the other "
                                   "(3 passes after the timed region); 'phases' below = the same events inside the timed region, "
                                   "where the duplex stream overlaps the McCaskill stream",
                         "kernels": kernels,
                         "whole_path": {"alg_GB_per_pair": b["total"] / 1e9 / batch,
                                        # every DP table written once and read once (SURVEY 8d's "compulsory bytes")
                                        "compulsory_GB_per_pair": compulsory_bytes(pairs, vienna, cofold) / 1e9 / batch,
                                        "achieved_GBs": b["total"] / 1e9 / (ms_mean[3] / 1e3),
                                        "frac": b["total"] / 1e9 / (ms_mean[3] / 1e3) / HBM_PEAK_GBS},
                         "phases": phases},
        }
        if world == 1:
            # PCIe-inclusive rates (never `value`): host strings in -> results on the host, for the record in DESIGN.md
            t1 = time.perf_counter()
            for _ in range(2):
                ctx.batch_upload(pairs); ctx.batch_compute(); ctx.batch_results_all()
            dense = 2 * batch / (time.perf_counter() - t1)
            t1 = time.perf_counter()
            for _ in range(2):
                ctx.batch_upload(pairs); ctx.batch_compute()
                for which, th in ((0, 0.5), (1, 0.5), (2, 0.1), (3, 0.003), (4, 0.003)):   # cmdline.c default thresholds
                    ctx.batch_candidates_all(which, th)
            sparse = 2 * batch / (time.perf_counter() - t1)
            line["pcie_inclusive"] = {"dense_results_pairs_per_s": dense, "threshold_candidates_pairs_per_s": sparse}
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
    ctx.close()


if __name__ == "__main__":
    main()
