"""Host logic of the multi-GPU shard: independent units, no data-path collective.

The z-score loop (/root/reference/src/ractip.cpp:1638-1657) is embarrassingly parallel over
iterations; each rank owns a contiguous block of iterations, computes their DPs on its own GPU,
and the per-iteration scalars are gathered once (RCCL on GPUs, gloo in the CPU tests) and
accumulated on rank 0 in ITERATION order, in float, as the reference does (:1655-1663).
"""
import ctypes
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
_zs = None


def _zscore_lib():
    global _zs
    if _zs is None:
        path = os.path.join(PKG, "host", "libractip_zscore.so")
        if not os.path.exists(path):
            raise RuntimeError(path + " is missing: run `python -m ractip_amd.build`")
        L = ctypes.CDLL(path)
        L.rh_zscore_shuffles.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_uint, ctypes.c_char_p, ctypes.c_char_p]
        L.rh_zscore_shuffles.restype = ctypes.c_int
        L.rh_zscore_from_energies.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float]
        L.rh_zscore_from_energies.restype = ctypes.c_float
        _zs = L
    return _zs


def zscore_shuffles(s1, s2, mode, num, seed):
    """All (s1', s2') pairs of the reference's z-score loop for --zscore=mode --seed=seed."""
    n1, n2 = len(s1), len(s2)
    b1 = ctypes.create_string_buffer(num * (n1 + 1))
    b2 = ctypes.create_string_buffer(num * (n2 + 1))
    if _zscore_lib().rh_zscore_shuffles(s1.encode(), s2.encode(), mode, num, seed, b1, b2) != 0:
        raise ValueError("zscore mode must be 1, 2 or 12")
    r1, r2 = b1.raw, b2.raw
    return [(r1[k * (n1 + 1):k * (n1 + 1) + n1].decode(), r2[k * (n2 + 1):k * (n2 + 1) + n2].decode())
            for k in range(num)]


def shard_bounds(num, rank, world):
    """Contiguous block [lo, hi) of `num` units owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(num, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_in_order(local, num, dist=None, device=None):
    """All ranks' per-unit rows, concatenated in unit order.  `local`: float array [hi-lo, k].
    One collective (all_gather of equal-sized, zero-padded blocks); identity when dist is None."""
    local = np.ascontiguousarray(local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = local.shape[1]
    cap = (num + world - 1) // world
    buf = torch.zeros((cap, width), dtype=torch.from_numpy(local).dtype)
    buf[:local.shape[0]] = torch.from_numpy(local)
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(num, r, world)
        rows.append(out[r][:hi - lo].cpu().numpy())
    return np.concatenate(rows, axis=0)


def zscore_from_energies(energies, e_native):
    """(e - mean)/sqrt(var) with the reference's float accumulation order (ractip.cpp:1655-1669)."""
    ee = np.ascontiguousarray(energies, dtype=np.float32)
    return float(_zscore_lib().rh_zscore_from_energies(ee.ctypes.data, ee.size, ctypes.c_float(e_native)))
