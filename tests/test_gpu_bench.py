"""GPU suite: bench.py's contract -- one JSON line with the required fields at N=1, and the multi-rank path
(torch.distributed.run, shard + gather + max-over-ranks) rehearsed with 2 ranks sharing the one GPU (gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def last_json(stdout):
    return json.loads([l for l in stdout.split("\n") if l.startswith("{")][-1])


def test_single_gpu_line(hotlib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "120", "--batch", "8", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "pairs/s" and d["value"] > 0
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "fp64") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert "workload" in d["config"] and "model" in d["config"] and d["config"]["model"] is None
    # the roofline kernel is a real kernel name with its own live, event-timed launch duration
    assert rf["kernel"] == rf["sweeps"][rf["sweep"]]["kernel"] and rf["avg_launch_us"] > 0 and rf["alg_bytes_per_launch"] > 0
    assert rf["kernel"].startswith(("lin_", "dxl_")) and rf["launches_per_step"] >= 1
    # `achieved` / `frac` are counter traffic over the live launch duration when a profile of these kernel sources exists (a fraction
    # that can fail: <= 1) and null otherwise -- never the algorithmic figure, which is reported under alg_* and may exceed 1
    alg = rf["alg_bytes_per_launch"] / 1e9 / (rf["avg_launch_us"] / 1e6)
    assert abs(rf["alg_GBs"] - alg) < 1e-6 * alg and abs(rf["alg_over_peak"] - alg / 8000.0) < 1e-9
    if rf["traffic"] is None:
        assert rf["frac"] is None and rf["achieved"] is None and rf["traffic_note"]
        assert rf["sweep_level"]["frac"] is None and rf["whole_path"]["hbm_frac"] is None
    else:
        assert rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
        assert rf["sweep_level"]["frac"] <= 1.0 and rf["whole_path"]["hbm_frac"] <= 1.0 and rf["whole_path"]["traffic_over_compulsory"] > 0
    assert rf["frac"] is None or rf["frac"] <= 1.0
    assert 0 < rf["fp64_frac"] < 1.0 and 0 < rf["whole_path"]["fp64_frac"] < 1.0
    # value is the end-to-end quantity (upload -> compute -> 5 scans on the host); the device-resident rate sits beside it
    assert d["device_resident_pairs_per_s"] > 0 and "rh_batch_upload" in d["config"]["step"]
    assert d["pcie_inclusive"]["dense_results_pairs_per_s"] > 0


def test_traffic_entries_are_bound_to_the_kernel_sources():
    """profiles/pmc_traffic.json entries carry the hash of the kernel sources they were measured on; bench.py's helper
    reproduces it, so a stale entry reads as `traffic: null` instead of a number from other kernels."""
    sys.path.insert(0, ROOT)
    import bench
    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    stamped = [k for k, v in doc.items() if isinstance(v, dict) and "source_hash" in v]
    for k in stamped:
        assert len(doc[k]["source_hash"]) == 16
    # an entry of the current format lists, per kernel, calibrated bytes per dispatch and dispatches per step
    for k in stamped:
        for name, v in doc[k].get("kernels", {}).items():
            assert v["bytes_per_dispatch"] >= 0 and v["dispatches_per_step"] > 0, (k, name)
    assert bench.sweep_of("void rh::lin_pack_tiles<0>") == "inside" and bench.sweep_of("void rh::lin_far2_outside") == "outside"
    assert bench.sweep_of("rh::dxl_strip8") == "duplex" and bench.sweep_of("cand_count_all") == "other"


def test_zscore_strong_scaling_line(hotlib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "zscore", "--total", "64", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["total_pairs_per_step"] == 64 and d["value"] > 0


def test_vienna_model_line(hotlib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "vienna", "--n", "100", "--batch", "4",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["value"] > 0 and "Vienna-BL" in d["config"]["scoring"] and d["roofline"]["kernel"].startswith(("vlin_", "dxv_"))
    assert d["roofline"]["frac"] is None or d["roofline"]["frac"] <= 1.0   # never an algorithmic figure under `frac`


def test_two_ranks_on_one_gpu_gloo(hotlib):
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--seqlen", "120",
           "--batch", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["pairs_per_gpu_per_step"] == 8


def test_two_ranks_strong_scaling_gloo(hotlib):
    """BASELINE config 5 as stated: a FIXED number of shuffles split over the ranks in contiguous iteration blocks."""
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "zscore",
           "--total", "51", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_pairs_per_step"] == 51
    assert d["config"]["pairs_per_gpu_per_step"] == 26 and d["value"] > 0
