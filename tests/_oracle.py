"""ctypes access to the TEST-ONLY oracle (oracle/libcf_oracle.so) and, when it has
been built in this tree, the reference engines (oracle/_ref/libref_contrafold.so)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
PARAMS = os.path.join(ROOT, "ractip_amd", "data", "contrafold_complementary.params")
NEG = -2e20


def tri_size(n):
    return (n + 1) * (n + 2) // 2


def tri_offset(n, i):
    return i * (2 * (n + 1) - i - 1) // 2


class Oracle:
    def __init__(self):
        so = os.path.join(ORACLE_DIR, "libcf_oracle.so")
        src = os.path.join(ORACLE_DIR, "cf_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        L = ctypes.CDLL(so)
        vp, cp, ci = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int
        L.cfo_load_params.restype = vp
        L.cfo_load_params.argtypes = [cp]
        L.cfo_inference.restype = ctypes.c_double
        L.cfo_inference.argtypes = [vp, cp, ci, vp, vp, vp]
        L.cfo_duplex.restype = None
        L.cfo_duplex.argtypes = [vp, cp, ci, cp, ci, vp, vp, vp, vp]
        L.cfo_up_float.restype = None
        L.cfo_up_float.argtypes = [ci, vp, vp]
        L.cfo_count_enable.argtypes = [ci]
        L.cfo_count_loads.restype = ctypes.c_ulonglong
        L.cfo_count_stores.restype = ctypes.c_ulonglong
        self.L = L
        self.m = L.cfo_load_params(PARAMS.encode())
        assert self.m, "oracle could not load " + PARAMS

    def inference(self, seq, tables=False):
        n = len(seq)
        T = tri_size(n)
        post = np.zeros(T)
        f5 = np.zeros(2 * (n + 1))
        tabs = np.zeros(6 * T) if tables else None
        z = self.L.cfo_inference(self.m, seq.encode(), n, post.ctypes.data,
                                 tabs.ctypes.data if tables else None, f5.ctypes.data)
        out = dict(logZ=z, post=post, f5=f5)
        if tables:
            out["tables"] = tabs
        return out

    def duplex(self, s1, s2):
        S = (len(s1) + 1) * (len(s2) + 1)
        post, ins, outs, z2 = np.zeros(S), np.zeros(S), np.zeros(S), np.zeros(2)
        self.L.cfo_duplex(self.m, s1.encode(), len(s1), s2.encode(), len(s2), post.ctypes.data,
                          ins.ctypes.data, outs.ctypes.data, z2.ctypes.data)
        shape = (len(s1) + 1, len(s2) + 1)
        return dict(logZ2=z2, post=post.reshape(shape), inside=ins.reshape(shape), outside=outs.reshape(shape))

    def up_float(self, n, bp_float32):
        up = np.zeros(n, dtype=np.float32)
        bp = np.ascontiguousarray(bp_float32, dtype=np.float32)
        self.L.cfo_up_float(n, bp.ctypes.data, up.ctypes.data)
        return up

    def count(self, fn, *args):
        """Algorithmic table loads/stores of one oracle call."""
        self.L.cfo_count_reset()
        self.L.cfo_count_enable(1)
        try:
            fn(*args)
        finally:
            self.L.cfo_count_enable(0)
        return self.L.cfo_count_loads(), self.L.cfo_count_stores()


class Reference:
    """The reference's own engines; only available where oracle/_ref was built."""

    def __init__(self):
        so = os.path.join(ORACLE_DIR, "_ref", "libref_contrafold.so")
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        L = ctypes.CDLL(so)
        L.ref_inference.restype = ctypes.c_double
        L.ref_inference.argtypes = [ctypes.c_char_p, ctypes.c_int] + [ctypes.c_void_p] * 3
        L.ref_duplex.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        self.L = L

    def inference(self, seq, use_float=False):
        n = len(seq)
        post = np.zeros(tri_size(n))
        z = self.L.ref_inference(seq.encode(), int(use_float), post.ctypes.data, None, None)
        return dict(logZ=z, post=post)

    def duplex(self, s1, s2, use_float=False):
        S = (len(s1) + 1) * (len(s2) + 1)
        post, z2 = np.zeros(S), np.zeros(2)
        self.L.ref_duplex(s1.encode(), s2.encode(), int(use_float), post.ctypes.data, None, None, z2.ctypes.data)
        return dict(logZ2=z2, post=post.reshape(len(s1) + 1, len(s2) + 1))


def assert_prob_close(got, ref, rel=1e-6, abs_floor=1e-12, what=""):
    """rel. err <= 1e-6 where |ref| > 1e-12, else abs. err <= 1e-12."""
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    assert got.shape == ref.shape, what
    big = np.abs(ref) > abs_floor
    if big.any():
        r = np.abs(got[big] - ref[big]) / np.abs(ref[big])
        assert r.max() <= rel, "%s: max rel err %.3e at %d (got %r ref %r)" % (
            what, r.max(), np.flatnonzero(big)[r.argmax()], got[big][r.argmax()], ref[big][r.argmax()])
    if (~big).any():
        a = np.abs(got[~big] - ref[~big])
        assert a.max() <= abs_floor, "%s: max abs err %.3e below the floor" % (what, a.max())


def assert_log_close(got, ref, tol=1e-9, what=""):
    """log-space tables: same -inf pattern, finite entries within tol (absolute, log units)."""
    got = np.asarray(got, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    gi, ri = got < NEG / 2, ref < NEG / 2
    assert (gi == ri).all(), "%s: -inf pattern differs at %s" % (what, np.flatnonzero(gi != ri)[:8])
    if (~ri).any():
        d = np.abs(got[~ri] - ref[~ri])
        assert d.max() <= tol, "%s: max |dlog| %.3e" % (what, d.max())


def ractip_constraint(structure, n):
    """The translation RactIP::rnafold applies to the FASTA structure line before pf_fold (src/ractip.cpp:271-287):
    '[', ']' and 'e' become 'x', everything else is passed through; missing positions are '.'."""
    c = ["."] * n
    for k, ch in enumerate(structure[:n]):
        c[k] = "x" if ch in "[]e" else ch
    return "".join(c)


def constraint_mask(constraint, n, turn=3):
    """Allowed-pair mask of ViennaRNA-1.8 pf_fold under fold_constrained (make_ptypes): (n+1)x(n+1) bytes, 1-based.
    'x' never pairs; '<' pairs only with a later letter; '>' only with an earlier one; a matched '(' ')' is kept and every
    pair inconsistent with it is removed; '|' and '.' do not restrict the partition function."""
    m = np.ones((n + 1, n + 1), dtype=np.uint8)
    m[0, :] = m[:, 0] = 0
    stack = []
    for j in range(1, n + 1):
        ch = constraint[j - 1] if j - 1 < len(constraint) else "."
        if ch == "x":
            m[:, j] = 0
            m[j, :] = 0
        elif ch in "(<":
            if ch == "(":
                stack.append(j)
            m[1:j, j] = 0          # j does not pair upstream
        elif ch in ")>":
            if ch == ")":
                i = stack.pop()
                keep = m[i, j]
                m[i:j + 1, j:n + 1] = 0
                m[1:i + 1, i:j + 1] = 0
                m[i, j] = keep
            m[j, j + 1:] = 0       # j does not pair downstream
    return np.ascontiguousarray(np.triu(m, 1))


class ViennaOracle:
    """BL*/ViennaRNA-1.8-semantics pf_duplex restatement (oracle/vienna_oracle.c) -- PARITY UNPINNED."""

    def __init__(self):
        so = os.path.join(ORACLE_DIR, "libvienna_oracle.so")
        src = os.path.join(ORACLE_DIR, "vienna_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        L = ctypes.CDLL(so)
        L.vo_load.restype = ctypes.c_void_p
        L.vo_load.argtypes = [ctypes.c_char_p]
        L.vo_pf_duplex.restype = ctypes.c_double
        L.vo_pf_duplex.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        L.vo_bruteforce.restype = ctypes.c_double
        L.vo_bruteforce.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
        L.vo_mccaskill.restype = ctypes.c_double
        L.vo_mccaskill.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vo_fold_bruteforce.restype = ctypes.c_double
        L.vo_fold_bruteforce.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.vo_mccaskill_cut.restype = ctypes.c_double
        L.vo_mccaskill_cut.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.vo_fold_bruteforce_cut.restype = ctypes.c_double
        L.vo_fold_bruteforce_cut.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.vo_set_allow_mask.argtypes = [ctypes.c_void_p]
        L.vo_set_allow_mask.restype = None
        self.L = L
        self.m = L.vo_load(os.path.join(ROOT, "ractip_amd", "data", "vienna_bl_star.params").encode())
        assert self.m

    def mccaskill(self, seq, max_w=0, tables=False, constraint=None):
        """pf_fold / pf_unstru semantics: logZ, bp (triangular, reference layout), up[n][max_w].
        constraint: a ViennaRNA constraint string (see constraint_mask) or None."""
        n = len(seq)
        mask = constraint_mask(constraint, n) if constraint is not None else None
        self.L.vo_set_allow_mask(mask.ctypes.data if mask is not None else None)
        try:
            return self._mccaskill(seq, max_w, tables)
        finally:
            self.L.vo_set_allow_mask(None)

    def _mccaskill(self, seq, max_w, tables):
        n = len(seq)
        post = np.zeros(tri_size(n))
        zo = ctypes.c_double()
        up = np.zeros((n, max_w)) if max_w else None
        tabs = np.zeros(8 * tri_size(n)) if tables else None
        f5 = np.zeros(2 * (n + 1)) if tables else None
        z = self.L.vo_mccaskill(self.m, seq.encode(), n, post.ctypes.data, ctypes.byref(zo),
                                up.ctypes.data if max_w else None, max_w,
                                tabs.ctypes.data if tables else None, f5.ctypes.data if tables else None)
        return dict(logZ=z, logZ_out=zo.value, post=post, up=up, tables=tabs, f5=f5)

    def cofold(self, s1, s2, bruteforce=False, constraint=None):
        mask = constraint_mask(constraint, len(s1) + len(s2)) if constraint is not None else None
        self.L.vo_set_allow_mask(mask.ctypes.data if mask is not None else None)
        try:
            return self._cofold(s1, s2, bruteforce)
        finally:
            self.L.vo_set_allow_mask(None)

    def _cofold(self, s1, s2, bruteforce=False):
        """co_pf_fold semantics on s1+s2 (cut after s1): logZ of the two-molecule ensemble, the full pair matrix of the
        concatenation (triangular, reference layout) and the intermolecular block hp[i][j] = P(s1[i] pairs s2[j]),
        (n1+1) x (n2+1), 1-based, the layout of src/ractip.cpp:451-454."""
        n1, n2 = len(s1), len(s2)
        n = n1 + n2
        post = np.zeros(tri_size(n))
        zo = ctypes.c_double()
        seq = (s1 + s2).encode()
        if bruteforce:
            z = self.L.vo_fold_bruteforce_cut(self.m, seq, n, n1, post.ctypes.data, None, 0)
            zo.value = z
        else:
            z = self.L.vo_mccaskill_cut(self.m, seq, n, n1, post.ctypes.data, ctypes.byref(zo), None, 0, None, None)
        hp = np.zeros((n1 + 1, n2 + 1))
        for i in range(1, n1 + 1):
            o = tri_offset(n, i)
            hp[i, 1:] = post[o + n1 + 1:o + n + 1]
        return dict(logZ=z, logZ_out=zo.value, post=post, hp=hp)

    def fold_bruteforce(self, seq, max_w=0, constraint=None):
        mask = constraint_mask(constraint, len(seq)) if constraint is not None else None
        self.L.vo_set_allow_mask(mask.ctypes.data if mask is not None else None)
        try:
            return self._fold_bruteforce(seq, max_w)
        finally:
            self.L.vo_set_allow_mask(None)

    def _fold_bruteforce(self, seq, max_w):
        n = len(seq)
        post = np.zeros(tri_size(n))
        up = np.zeros((n, max_w)) if max_w else None
        z = self.L.vo_fold_bruteforce(self.m, seq.encode(), n, post.ctypes.data, up.ctypes.data if max_w else None, max_w)
        return dict(logZ=z, post=post, up=up)

    def pf_duplex(self, s1, s2):
        pr = np.zeros((len(s1) + 1, len(s2) + 1))
        ebk = ctypes.c_double()
        z = self.L.vo_pf_duplex(self.m, s1.encode(), len(s1), s2.encode(), len(s2), pr.ctypes.data, None, None, ctypes.byref(ebk))
        return dict(logZ=z, logZ_bk=ebk.value, pr=pr)

    def bruteforce(self, s1, s2):
        pr = np.zeros((len(s1) + 1, len(s2) + 1))
        z = self.L.vo_bruteforce(self.m, s1.encode(), len(s1), s2.encode(), len(s2), pr.ctypes.data)
        return dict(logZ=z, pr=pr)
