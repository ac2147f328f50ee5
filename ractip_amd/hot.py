"""ctypes binding of include/ractip_hot.h (the C ABI of libractip_hot.so).

Thin by design: every method is one C call.  There is NO fallback: if the HIP
library is missing or no GPU is present, creation raises.
"""
import ctypes
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RACTIP_HOT_LIB") or os.path.join(PKG, "libractip_hot.so")   # override: tuning builds (tools/build_variant.py)

RH_MODEL_CONTRAFOLD = 0
RH_MODEL_VIENNA_BL = 1

EXPORTS = [
    "rh_create", "rh_destroy", "rh_last_error", "rh_set_mode", "rh_last_path", "rh_bpp", "rh_unpaired", "rh_fold", "rh_duplex",
    "rh_batch_upload", "rh_batch_compute", "rh_batch_results", "rh_batch_candidates",
    "rh_batch_timings", "rh_batch_device_views", "rh_batch_logz", "rh_batch_candidates_all", "rh_batch_layout",
    "rh_batch_results_all", "rh_set_max_w", "rh_get_max_w", "rh_set_overlap", "rh_batch_kernels", "rh_set_hybrid", "rh_last_hybrid_path", "rh_fold_constrained", "rh_cofold_constrained",
    "rh_host_alloc", "rh_host_free", "rh_batch_fallbacks", "rh_create_vienna", "rh_vienna_semantics", "rh_set_scale_memory", "rh_set_kernel_timing", "rh_kernel_times",
    "rh_debug_vienna_cell", "rh_debug_vienna_value",   # loader inspection (host only; used by the CPU tests of the loader)
]


class RhError(RuntimeError):
    pass


class Cand(ctypes.Structure):
    _fields_ = [("i", ctypes.c_int), ("j", ctypes.c_int), ("p", ctypes.c_float)]


_lib = None


def load_library():
    """dlopen libractip_hot.so and declare the prototypes (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RhError("%s is missing: run `python -m ractip_amd.build` (or __graft_entry__.build()) first; "
                      "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, ci, dp = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)
    L.rh_create.restype = vp
    L.rh_create_vienna.restype = vp
    L.rh_create_vienna.argtypes = [ci, cp, ci, cp, ci]
    L.rh_vienna_semantics.argtypes = [vp]
    L.rh_create.argtypes = [ci, ci, cp]
    L.rh_destroy.restype = None
    L.rh_destroy.argtypes = [vp]
    L.rh_last_error.restype = cp
    L.rh_last_error.argtypes = [vp]
    L.rh_set_mode.argtypes = [vp, ci]
    L.rh_set_mode.restype = ci
    L.rh_last_path.argtypes = [vp]
    L.rh_set_overlap.argtypes = [vp, ci]
    L.rh_set_scale_memory.argtypes = [vp, ci]
    L.rh_set_kernel_timing.argtypes = [vp, ci]
    L.rh_set_kernel_timing.restype = ci
    L.rh_kernel_times.argtypes = [vp, vp, vp]
    L.rh_kernel_times.restype = ci
    L.rh_set_scale_memory.restype = ci
    L.rh_last_hybrid_path.argtypes = [vp]
    L.rh_last_hybrid_path.restype = ci
    L.rh_set_hybrid.argtypes = [vp, ci]
    L.rh_set_hybrid.restype = ci
    L.rh_batch_kernels.argtypes = [vp, vp, vp, vp]
    L.rh_batch_kernels.restype = ci
    L.rh_set_overlap.restype = ci
    L.rh_set_max_w.argtypes = [vp, ci]
    L.rh_set_max_w.restype = ci
    L.rh_get_max_w.argtypes = [vp]
    L.rh_get_max_w.restype = ci
    L.rh_last_path.restype = ci
    L.rh_bpp.argtypes = [vp, cp, ci, cp, vp, vp]
    L.rh_unpaired.argtypes = [vp, cp, ci, ci, vp]
    L.rh_fold.argtypes = [vp, cp, ci, vp, vp, vp]
    L.rh_fold_constrained.argtypes = [vp, cp, ci, cp, vp, vp, vp]
    L.rh_fold_constrained.restype = ci
    L.rh_cofold_constrained.argtypes = [vp, cp, ci, cp, ci, cp, vp, vp]
    L.rh_cofold_constrained.restype = ci
    L.rh_duplex.argtypes = [vp, cp, ci, cp, ci, vp, vp]
    L.rh_batch_upload.argtypes = [vp, ci, ctypes.POINTER(cp), ctypes.POINTER(ci), ctypes.POINTER(cp), ctypes.POINTER(ci)]
    L.rh_batch_compute.argtypes = [vp]
    L.rh_batch_results.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    L.rh_batch_candidates.argtypes = [vp, ci, ci, ctypes.c_float, vp, ci]
    L.rh_batch_candidates_all.argtypes = [vp, ci, ctypes.c_float, vp, ci, vp]
    L.rh_batch_layout.argtypes = [vp, vp, vp, vp, vp]
    L.rh_batch_results_all.argtypes = [vp, vp, vp, vp, vp]
    L.rh_batch_timings.argtypes = [vp, vp, vp]
    L.rh_batch_logz.argtypes = [vp, vp]
    L.rh_batch_device_views.argtypes = [vp, vp, vp, vp, vp, vp]
    L.rh_batch_fallbacks.argtypes = [vp, ci, vp, ci]
    L.rh_batch_fallbacks.restype = ci
    L.rh_host_alloc.restype = vp
    L.rh_host_alloc.argtypes = [vp, ctypes.c_size_t]
    L.rh_host_free.restype = None
    L.rh_host_free.argtypes = [vp, vp]
    for f in ("rh_bpp", "rh_unpaired", "rh_fold", "rh_duplex", "rh_batch_upload", "rh_batch_compute", "rh_batch_results",
              "rh_batch_candidates", "rh_batch_timings", "rh_batch_device_views", "rh_batch_logz", "rh_batch_candidates_all",
              "rh_batch_layout", "rh_batch_results_all"):
        getattr(L, f).restype = ci
    _lib = L
    return L


def tri_size(n):
    return (n + 1) * (n + 2) // 2


def tri_offset(n, i):
    return i * (2 * (n + 1) - i - 1) // 2


class Context:
    """One rh_ctx: one GPU, one host thread."""

    def __init__(self, device=0, model=RH_MODEL_CONTRAFOLD, param_file=None, vienna=None):
        """vienna = dict(defaults_file=None, use_bl_param=True, semantics=0): rh_create_vienna (Vienna model only) -- the
        tables as RactIP::run installs them (library defaults -> BL* -> -P file) and the 1.8 / 2.x loop-energy semantics."""
        self.L = load_library()
        enc = lambda x: x.encode() if x else None
        if vienna is not None:
            if model != RH_MODEL_VIENNA_BL:
                raise RhError("vienna= applies to RH_MODEL_VIENNA_BL")
            self.h = self.L.rh_create_vienna(device, enc(vienna.get("defaults_file")), 1 if vienna.get("use_bl_param", True) else 0,
                                             enc(param_file), int(vienna.get("semantics", 0)))
        else:
            self.h = self.L.rh_create(device, model, enc(param_file))
        if not self.h:
            raise RhError("rh_create failed: %s" % self.L.rh_last_error(None).decode())
        self._pairs = []

    def vienna_semantics(self):
        """1 = ViennaRNA-1.8 loop energies, 2 = ViennaRNA-2.x (0: CONTRAfold model)."""
        return self.L.rh_vienna_semantics(self.h)

    def close(self):
        if getattr(self, "h", None):
            for ptr in getattr(self, "_pinned", []):
                self.L.rh_host_free(self.h, ptr)
            self._pinned = []
            self.L.rh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise RhError("ractip_hot error %d: %s" % (rc, self.L.rh_last_error(self.h).decode()))
        return rc

    def set_mode(self, mode):
        """0 = auto (linear, log-space fallback), 1 = log-space, 2 = linear only."""
        self._check(self.L.rh_set_mode(self.h, mode))

    def last_path(self):
        return self.L.rh_last_path(self.h)

    def set_hybrid(self, cofold):
        """Vienna-BL model: hp from pf_duplex (False, default) or from the two-molecule ensemble of co_pf_fold (True)."""
        self._check(self.L.rh_set_hybrid(self.h, 1 if cofold else 0))

    def last_hybrid_path(self):
        return self.L.rh_last_hybrid_path(self.h)

    def set_max_w(self, max_w):
        self._check(self.L.rh_set_max_w(self.h, max_w))

    @property
    def max_w(self):
        return self.L.rh_get_max_w(self.h)

    # ---- single-problem calls
    def bpp(self, seq, constraint=None):
        n = len(seq)
        bp = np.zeros(tri_size(n))
        z = ctypes.c_double()
        self._check(self.L.rh_bpp(self.h, seq.encode(), n, constraint.encode() if constraint is not None else None,
                                  bp.ctypes.data, ctypes.addressof(z)))
        return bp, z.value

    def unpaired(self, seq, max_w=1):
        n = len(seq)
        up = np.zeros(n * max_w)
        self._check(self.L.rh_unpaired(self.h, seq.encode(), n, max_w, up.ctypes.data))
        return up.reshape(n, max_w)

    def fold(self, seq, constraint=None):
        n = len(seq)
        w = self.max_w
        bp, up, z = np.zeros(tri_size(n)), np.zeros(n * w), ctypes.c_double()
        if constraint is not None:
            self._check(self.L.rh_fold_constrained(self.h, seq.encode(), n, constraint.encode(), bp.ctypes.data, up.ctypes.data,
                                                   ctypes.addressof(z)))
        else:
            self._check(self.L.rh_fold(self.h, seq.encode(), n, bp.ctypes.data, up.ctypes.data, ctypes.addressof(z)))
        return bp, (up if w == 1 else up.reshape(n, w)), z.value

    def duplex(self, s1, s2):
        hp = np.zeros((len(s1) + 1, len(s2) + 1))
        z = ctypes.c_double()
        self._check(self.L.rh_duplex(self.h, s1.encode(), len(s1), s2.encode(), len(s2), hp.ctypes.data, ctypes.addressof(z)))
        return hp, z.value

    def cofold(self, s1, s2, constraint=None):
        """hp and log Z of the two-molecule ensemble, optionally under a constraint over s1+s2 (Vienna-BL model)."""
        hp = np.zeros((len(s1) + 1, len(s2) + 1))
        z = ctypes.c_double()
        self._check(self.L.rh_cofold_constrained(self.h, s1.encode(), len(s1), s2.encode(), len(s2),
                                                 constraint.encode() if constraint is not None else None, hp.ctypes.data, ctypes.addressof(z)))
        return hp, z.value

    # ---- batched calls
    def batch_upload(self, pairs):
        np_ = len(pairs)
        a = (ctypes.c_char_p * np_)(*[p[0].encode() for p in pairs])
        b = (ctypes.c_char_p * np_)(*[p[1].encode() for p in pairs])
        na = (ctypes.c_int * np_)(*[len(p[0]) for p in pairs])
        nb = (ctypes.c_int * np_)(*[len(p[1]) for p in pairs])
        self._check(self.L.rh_batch_upload(self.h, np_, a, na, b, nb))
        self._pairs = [(len(p[0]), len(p[1])) for p in pairs]

    def batch_compute(self):
        self._check(self.L.rh_batch_compute(self.h))

    def batch_results(self, p):
        n1, n2 = self._pairs[p]
        bp1, bp2 = np.zeros(tri_size(n1)), np.zeros(tri_size(n2))
        w = self.max_w
        up1, up2 = np.zeros(n1 * w), np.zeros(n2 * w)
        hp = np.zeros((n1 + 1, n2 + 1))
        z3 = np.zeros(3)
        self._check(self.L.rh_batch_results(self.h, p, bp1.ctypes.data, bp2.ctypes.data, up1.ctypes.data,
                                            up2.ctypes.data, hp.ctypes.data, z3.ctypes.data))
        if w > 1:
            up1, up2 = up1.reshape(n1, w), up2.reshape(n2, w)
        return dict(bp1=bp1, bp2=bp2, up1=up1, up2=up2, hp=hp, logZ=z3)

    def batch_logz(self):
        out = np.zeros((len(self._pairs), 3))
        self._check(self.L.rh_batch_logz(self.h, out.ctypes.data))
        return out

    def batch_candidates(self, p, which, threshold, cap=1 << 20):
        buf = (Cand * cap)()
        k = self._check(self.L.rh_batch_candidates(self.h, p, which, ctypes.c_float(threshold), buf, cap))
        self.last_candidate_count = k
        return [(buf[t].i, buf[t].j, buf[t].p) for t in range(min(k, cap))]

    CAND_DTYPE = np.dtype([("i", np.int32), ("j", np.int32), ("p", np.float32)])

    def batch_candidates_all(self, which, threshold, cap=1 << 22):
        """Candidates of every pair: (records, first) -- a structured array (i, j, p) and the np+1 offsets
        such that pair k owns records[first[k]:first[k+1]].  No per-entry Python work."""
        if getattr(self, "_cand_buf", None) is None or self._cand_buf.size < cap:
            self._cand_buf = np.empty(cap, dtype=self.CAND_DTYPE)
        first = np.empty(len(self._pairs) + 1, dtype=np.int32)
        k = self._check(self.L.rh_batch_candidates_all(self.h, which, ctypes.c_float(threshold), self._cand_buf.ctypes.data,
                                                       cap, first.ctypes.data))
        if k > cap:
            raise RhError("candidate buffer too small: %d > %d" % (k, cap))
        return self._cand_buf[:k], first

    def batch_results_all(self):
        """Dense results of all pairs (three device-to-host copies), unpacked into per-pair dicts."""
        ts, hs = ctypes.c_size_t(), ctypes.c_size_t()
        uld, hld = ctypes.c_int(), ctypes.c_int()
        self._check(self.L.rh_batch_layout(self.h, ctypes.byref(ts), ctypes.byref(uld), ctypes.byref(hs), ctypes.byref(hld)))
        np_ = len(self._pairs)
        bp = np.empty((2 * np_, ts.value)); up = np.empty((2 * np_, uld.value)); hp = np.empty((np_, hs.value)); z = np.empty((np_, 3))
        self._check(self.L.rh_batch_results_all(self.h, bp.ctypes.data, up.ctypes.data, hp.ctypes.data, z.ctypes.data))
        out = []
        w = self.max_w
        for p, (n1, n2) in enumerate(self._pairs):
            h = hp[p][:(n1 + 1) * hld.value].reshape(n1 + 1, hld.value)[:, :n2 + 1]
            u1, u2 = up[2 * p][:n1 * w], up[2 * p + 1][:n2 * w]
            if w > 1:
                u1, u2 = u1.reshape(n1, w), u2.reshape(n2, w)
            out.append(dict(bp1=bp[2 * p][:tri_size(n1)], bp2=bp[2 * p + 1][:tri_size(n2)], up1=u1, up2=u2, hp=h, logZ=z[p]))
        return out

    def pinned_empty(self, shape, dtype=np.float64):
        """numpy array over page-locked host memory (rh_host_alloc).  The memory belongs to the context: it is released by
        pinned_free(array) or close(), and the array must not be touched afterwards."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self.L.rh_host_alloc(self.h, max(nbytes, 8))
        if not ptr:
            raise RhError("rh_host_alloc failed: %s" % self.L.rh_last_error(self.h).decode())
        self._pinned = getattr(self, "_pinned", []) + [ptr]
        buf = (ctypes.c_char * max(nbytes, 8)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def pinned_free(self, arr):
        """Release one pinned_empty array now (rh_host_free) instead of with the context."""
        ptr = arr.ctypes.data
        pinned = getattr(self, "_pinned", [])
        if ptr in pinned:
            pinned.remove(ptr)
            self.L.rh_host_free(self.h, ptr)

    def batch_results_all_into(self, bufs=None):
        """Dense results of all pairs in the padded device layout (rh_batch_layout), three copies, no unpacking.  `bufs` =
        (bp, up, hp, logz) arrays from a previous call are reused when the layout is unchanged; they are page-locked."""
        ts, hs = ctypes.c_size_t(), ctypes.c_size_t()
        uld, hld = ctypes.c_int(), ctypes.c_int()
        self._check(self.L.rh_batch_layout(self.h, ctypes.byref(ts), ctypes.byref(uld), ctypes.byref(hs), ctypes.byref(hld)))
        np_ = len(self._pairs)
        shapes = ((2 * np_, ts.value), (2 * np_, uld.value), (np_, hs.value), (np_, 3))
        if bufs is None or tuple(b.shape for b in bufs) != shapes:
            for b in bufs or ():          # the layout changed: the buffers this call replaces are released, not kept until close()
                self.pinned_free(b)
            bufs = tuple(self.pinned_empty(sh) for sh in shapes)
        self._check(self.L.rh_batch_results_all(self.h, *[b.ctypes.data for b in bufs]))
        return bufs

    def batch_fallbacks(self, which=0):
        """Indices of the sequences (which=0) / pairs (which=1) the last compute recomputed in log space; which=2: the sequences it
        recomputed on the linear kernels with another scale exponent; which=3: the pairs whose duplex sweeps it recomputed that way."""
        buf = (ctypes.c_int * max(1, 2 * len(self._pairs)))()
        k = self._check(self.L.rh_batch_fallbacks(self.h, which, buf, len(buf)))
        return [buf[t] for t in range(min(k, len(buf)))]

    def set_scale_memory(self, on):
        """True: the next batch starts on the scale exponent most of the last one needed (faster on streams of one kind of input;
        a sequence's bits then depend on the context's history).  Default off."""
        self._check(self.L.rh_set_scale_memory(self.h, 1 if on else 0))

    def kernel_class_times(self, cls, computes=1):
        """Average duration (us) and launches per compute of ONE class of sweep kernels (rh_set_kernel_timing: 0 inside sweep kernel,
        1 its block products + packing, 2 outside sweep kernel, 3 its block products, 4 duplex sweep kernel), measured live with HIP
        event pairs around every launch of that class over `computes` passes of the current batch, phases not overlapping."""
        self._check(self.L.rh_set_kernel_timing(self.h, cls))
        self._check(self.L.rh_set_overlap(self.h, 0))
        n_tot, ms_tot = 0, 0.0
        try:
            for _ in range(computes):
                self.batch_compute()
                n, ms = ctypes.c_int(), ctypes.c_double()
                self._check(self.L.rh_kernel_times(self.h, ctypes.byref(n), ctypes.byref(ms)))
                n_tot += n.value
                ms_tot += ms.value
        finally:
            self.L.rh_set_kernel_timing(self.h, -1)
            self.L.rh_set_overlap(self.h, 1)
        return (ms_tot * 1e3 / n_tot if n_tot else 0.0), n_tot / float(computes)

    def set_overlap(self, on):
        """False: phases run one after the other (isolated per-phase device times in batch_timings)."""
        self._check(self.L.rh_set_overlap(self.h, 1 if on else 0))

    def batch_kernels(self):
        """[(per-diagonal kernel, block-product kernel or '', block-product launches)] for inside, outside, duplex."""
        fine, far, nf = (ctypes.c_char_p * 3)(), (ctypes.c_char_p * 3)(), (ctypes.c_int * 3)()
        self._check(self.L.rh_batch_kernels(self.h, fine, far, nf))
        return [((fine[k] or b"").decode(), (far[k] or b"").decode(), nf[k]) for k in range(3)]

    def batch_timings(self):
        ms = (ctypes.c_double * 4)()
        nl = (ctypes.c_int * 3)()
        self._check(self.L.rh_batch_timings(self.h, ms, nl))
        return list(ms), list(nl)
