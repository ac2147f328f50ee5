"""Algorithmic bytes B_alg of the hot path.

B_alg = 8 B x (DP-table element loads + stores that the REFERENCE recurrences
perform), score-table gathers and sequence bytes excluded.  This module counts
them exactly from the pairability pattern of the actual input with vectorised
numpy (no DP is run); tests/test_balg.py pins it to the instrumented CPU oracle.

Reference loops being counted (/root/reference/src/contrafold):
  InferenceEngine.ipp  inside 3376-3717, outside 3751-4064, posterior 4516-4817
  DuplexEngine.ipp     inside 1023-1075, outside 1088-1141, posterior 1153-1160
"""
import numpy as np

_PAIR = np.zeros((5, 5), dtype=bool)
for _a, _b in ((0, 3), (3, 0), (1, 2), (2, 1), (2, 3), (3, 2)):
    _PAIR[_a, _b] = True
_CODE = {c: k for k, c in enumerate("ACGU")}
_CODE.update({c.lower(): k for c, k in list(_CODE.items())})


def _codes(seq):
    return np.array([_CODE.get(c, 4) for c in seq], dtype=np.int64)


def mccaskill_counts(seq):
    """(loads, stores) per phase for one sequence: dict inside/outside/posterior."""
    L = len(seq)
    s = np.full(L + 2, 4, dtype=np.int64)
    s[1:L + 1] = _codes(seq)
    A = _PAIR[s[:, None], s[None, :]]  # A[x][y]: letters x,y may pair (1-based), x<y enforced below
    A = np.triu(A, 1)
    k3 = (L + 1) * L * (L - 1) // 6
    # guardC cells (i,j): 0<i<=j<L, letters (i,j+1) pairable; inner pair letters (p+1,q) = (i+l1+1, j-l2)
    C = np.zeros((L + 2, L + 2), dtype=bool)  # C[i][j] = guardC
    if L >= 2:
        C[1:L, 1:L] = A[1:L, 2:L + 1]
        C = np.triu(C)  # j >= i
    n_c = int(C.sum())
    n_sb = 0
    for t in range(0, 31):
        for l1 in range(0, t + 1):
            l2 = t - l1
            # cells (i,j) with j-l2 >= i+l1+2 ; inner letters (i+l1+1, j-l2) pairable
            # shift: inner[i][j] = A[i+l1+1][j-l2]
            i0, i1 = 1, L - 1 - l1  # i+l1+1 <= L
            j0, j1 = 1 + l2, L - 1  # j-l2 >= 1
            if i1 < i0 or j1 < j0:
                continue
            cc = C[i0:i1 + 1, j0:j1 + 1]
            inner = A[i0 + l1 + 1:i1 + l1 + 2, j0 - l2:j1 - l2 + 1]
            ii = np.arange(i0, i1 + 1)[:, None]
            jj = np.arange(j0, j1 + 1)[None, :]
            ok = (jj - l2) >= (ii + l1 + 2)
            n_sb += int((cc & inner & ok).sum())
    # guardM cells: 0<i, i+2<=j, j<L
    n_m = (L - 3) * (L - 2) // 2 if L >= 3 else 0
    # guardM cells whose letters (i+1, j) are pairable
    n_mp = 0
    if L >= 3:
        Mx = np.zeros((L + 2, L + 2), dtype=bool)
        Mx[1:L - 1, 3:L] = A[2:L, 3:L]  # cell (i,j) -> A[i+1][j]
        ii = np.arange(L + 2)[:, None]
        jj = np.arange(L + 2)[None, :]
        n_mp = int((Mx & (jj >= ii + 2) & (ii >= 1) & (jj <= L - 1)).sum())
    n_pairs = int(A[1:L + 1, 1:L + 1].sum())  # pairable (k+1, j), all spans
    inside = (2 * k3 + n_sb + n_mp + 2 * n_m + 2 * n_pairs, n_c + 2 * n_m + L)
    outside = (L + 4 * n_pairs + (4 * n_m + n_mp) + n_c + n_sb + 4 * k3,
               L + 2 * n_pairs + (3 * n_m + n_mp) + n_sb + 2 * k3)
    posterior = (n_c + 2 * n_sb + 3 * n_mp + 3 * n_pairs, n_sb + n_mp + n_pairs)
    return dict(inside=inside, outside=outside, posterior=posterior)


def duplex_counts(s1, s2):
    """(loads, stores) per phase for one pair."""
    L1, L2 = len(s1), len(s2)
    a = np.full(L1 + 2, 4, dtype=np.int64)
    b = np.full(L2 + 2, 4, dtype=np.int64)
    a[1:L1 + 1] = _codes(s1)
    b[1:L2 + 1] = _codes(s2)
    A = _PAIR[a[:, None], b[None, :]]
    A[0, :] = A[:, 0] = False
    n_p = int(A.sum())
    n_in = 0
    for t in range(0, 29):
        for l1 in range(0, t + 1):
            l2 = t - l1
            # (i,j) with p = i-1-l1 >= 1, q = j+1+l2 <= L2
            i0, i1 = 2 + l1, L1
            j0, j1 = 1, L2 - 1 - l2
            if i1 < i0 or j1 < j0:
                continue
            n_in += int((A[i0:i1 + 1, j0:j1 + 1] & A[i0 - 1 - l1:i1 - l1, j0 + 1 + l2:j1 + l2 + 2]).sum())
    return dict(inside=(n_in, n_p), outside=(n_p + n_in, n_p + n_in), posterior=(2 * n_p, n_p))


def far_terms(n, bs=16):
    """How many k-terms of the O(n^3) sums the block-product kernels (mccaskill_far.hip) take over from the
    per-diagonal kernels, for one sequence of length n and block size bs: (inside FM2 terms, outside FMo terms,
    outside FM1o terms).  Mirrors the near/far split of lin_inside_diag / lin_outside_diag exactly: cell (i,j),
    I = i//bs, J = j//bs; inside: k in [(I+2)bs, (J-1)bs) when J-I >= 4; outside: FMo sources i' < (I-1)bs,
    FM1o sources j' >= (J+2)bs (cells with j-i >= 2 only)."""
    if n < 3 or bs <= 0:
        return 0, 0, 0
    i = np.arange(1, n)[:, None]
    j = np.arange(1, n)[None, :]
    valid = j >= i
    I, J = i // bs, j // bs
    k_in = np.where(valid & (J - I >= 4), (J - I - 3) * bs, 0).sum()
    gm = valid & (j - i >= 2)
    k_fmo = np.where(gm, np.maximum((I - 1) * bs - 1, 0), 0).sum()
    k_fm1o = np.where(gm, np.maximum(n - (J + 2) * bs, 0), 0).sum()
    return int(k_in), int(k_fmo), int(k_fm1o)


def pair_bytes_by_kernel(s1, s2, bs=16):
    """B_alg of one pair split by the kernel class that executes the accesses (rh_batch_kernel_times order):
    inside fine diagonals, inside block products, outside fine, outside block products, duplex sweep.  A k-term of
    the reference costs 2 table accesses in the inside sweep (InferenceEngine.ipp:3396-3403) and 6 in the outside
    sweep (two read-modify-writes and two operand loads, :4046-4064), 3 per updated table."""
    b = pair_bytes(s1, s2)
    far_in = far_out = 0
    for s in (s1, s2):
        k_in, k_fmo, k_fm1o = far_terms(len(s), bs)
        far_in += 8 * 2 * k_in
        far_out += 8 * 3 * (k_fmo + k_fm1o)
    return dict(inside=b["mc_inside"] - far_in, inside_far=far_in, outside=b["mc_outside"] - far_out, outside_far=far_out,
                duplex=b["duplex"], total=b["total"])


def pair_bytes(s1, s2):
    """B_alg of one (s1,s2) pair split by GPU kernel: inside sweep, outside sweep (+posterior), duplex."""
    tot = dict(mc_inside=0, mc_outside=0, duplex=0)
    for s in (s1, s2):
        c = mccaskill_counts(s)
        tot["mc_inside"] += 8 * sum(c["inside"])
        tot["mc_outside"] += 8 * (sum(c["outside"]) + sum(c["posterior"]))
    d = duplex_counts(s1, s2)
    tot["duplex"] = 8 * (sum(d["inside"]) + sum(d["outside"]) + sum(d["posterior"]))
    tot["total"] = tot["mc_inside"] + tot["mc_outside"] + tot["duplex"]
    return tot
