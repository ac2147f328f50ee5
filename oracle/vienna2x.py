"""oracle/vienna2x.py -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

CPU restatement, in plain Python, of what the product computes under RH_VIENNA_SEM_20 (and, for the pieces the two share,
RH_VIENNA_SEM_18):

  * a reader / writer of ViennaRNA parameter files ("## RNAfold parameter file v2.0"), written independently of
    ractip_amd/csrc/vienna_loader.cpp (different tokenizer, numpy tables) so that the two can be checked against each other;
  * the ViennaRNA-2.x loop energies the HAVE_VIENNA20 branch of /root/reference/src/pf_duplex.c:128-206 calls -- E_ExtLoop
    (:146,158,185,200) and E_IntLoop (:153,193) -- restated from the published ViennaRNA 2.x loop_energies.h
    (ViennaRNA is a third-party dependency of the reference, RNAlib2 >= 2.2.0 by README, not vendored, no pinned version);
  * pf_duplex_fw / pf_duplex_bk / pr_duplex exactly as pf_duplex.c:42-206 nests its loops;
  * brute-force enumeration of every duplex (chain of inter-molecular pairs) and of every secondary structure of a short
    sequence under the same energies, to pin the DPs (this one's and the GPU's) to the definition of the ensemble.

Nothing in the reference holds an output of this path, so all of it is "parity unpinned": it proves that the kernels
evaluate the stated energy model, not that the model is the one RNAlib2 implements.
"""
import itertools
import math

import numpy as np

INF = 1000000
MAXLOOP = 30
KT = (37.0 + 273.15) * 1.98717          # (temperature+K0)*GASCONST, pf_duplex.c:73
RTYPE = [0, 2, 1, 4, 3, 6, 5, 7]
PAIR = np.zeros((5, 5), dtype=int)      # A,C,G,U = 1..4: CG=1 GC=2 GU=3 UG=4 AU=5 UA=6
for (a, b), t in {(2, 3): 1, (3, 2): 2, (3, 4): 3, (4, 3): 4, (1, 4): 5, (4, 1): 6}.items():
    PAIR[a, b] = t
MISMATCH_SECTIONS = {"mismatch_hairpin": "mismatchH", "mismatch_interior": "mismatchI", "mismatch_interior_1n": "mismatch1nI",
                     "mismatch_interior_23": "mismatch23I", "mismatch_multi": "mismatchM", "mismatch_exterior": "mismatchExt"}


def encode(seq):
    """encode_sequence: A,C,G,U -> 1..4 (T as U), anything else 0; index 0 and n+1 are padding."""
    return [0] + ["ACGU".find(c if c != "T" else "U") + 1 for c in seq.upper()] + [0]


def empty_tables():
    T = {"stack": np.zeros((8, 8), int), "dangle5": np.zeros((8, 5), int), "dangle3": np.zeros((8, 5), int),
         "int11": np.zeros((8, 8, 5, 5), int), "int21": np.zeros((8, 8, 5, 5, 5), int), "int22": np.zeros((8, 8, 5, 5, 5, 5), int),
         "hairpin": np.zeros(31, int), "bulge": np.zeros(31, int), "interior": np.zeros(31, int),
         "ninio": 0, "max_ninio": 300, "ML_base": 0, "ML_closing": 0, "ML_intern": 0, "TerminalAU": 0, "DuplexInit": 410, "lxc": 107.856,
         "Tetraloops": {}, "Triloops": {}, "Hexaloops": {}, "v20": False}
    for name in MISMATCH_SECTIONS.values():
        T[name] = np.zeros((8, 5, 5), int)
    return T


def random_tables(seed):
    """synthetic tables: every entry the 2.x energy functions can reach gets its own random value"""
    rng = np.random.default_rng(seed)
    T = empty_tables()
    T["v20"] = True
    T["stack"][1:, 1:] = rng.integers(-350, -50, (7, 7))
    for name in MISMATCH_SECTIONS.values():
        T[name][1:] = rng.integers(-160, 60, (7, 5, 5))
    T["dangle5"][1:] = rng.integers(-90, 30, (7, 5))
    T["dangle3"][1:] = rng.integers(-110, 30, (7, 5))
    T["int11"][1:, 1:] = rng.integers(-100, 250, (7, 7, 5, 5))
    T["int21"][1:, 1:] = rng.integers(0, 400, (7, 7, 5, 5, 5))
    T["int22"][1:7, 1:7, 1:, 1:, 1:, 1:] = rng.integers(-50, 350, (6, 6, 4, 4, 4, 4))
    T["hairpin"][:3] = INF
    T["hairpin"][3:] = np.sort(rng.integers(350, 800, 28))
    T["bulge"][0] = INF
    T["bulge"][1:] = np.sort(rng.integers(250, 700, 30))
    T["interior"][:2] = INF
    T["interior"][2:] = np.sort(rng.integers(50, 450, 29))
    T["ninio"], T["max_ninio"] = int(rng.integers(30, 80)), 300
    T["ML_base"], T["ML_closing"], T["ML_intern"] = int(rng.integers(-10, 30)), int(rng.integers(200, 1000)), int(rng.integers(-120, 60))
    T["TerminalAU"], T["DuplexInit"], T["lxc"] = int(rng.integers(20, 90)), int(rng.integers(300, 500)), 107.856
    T["Tetraloops"] = {"GGGGAC": int(rng.integers(-300, 500)), "CGAAAG": int(rng.integers(-300, 500)), "UACGAG": int(rng.integers(100, 600))}
    T["Triloops"] = {"CAACG": int(rng.integers(400, 800)), "GUUAC": int(rng.integers(400, 800))}
    T["Hexaloops"] = {"ACAGUACU": int(rng.integers(100, 500)), "CCGAGAGG": int(rng.integers(100, 500))}
    return T


def write_par_v20(path, T):
    """a parameter file in the v2.0 layout: comments, enthalpy sections (to be skipped), INF and DEF tokens"""
    def tok(v):
        return "INF" if v >= INF else str(int(v))

    def rows(a, width):
        flat = [tok(v) for v in np.asarray(a).reshape(-1)]
        return "\n".join(" ".join("%6s" % x for x in flat[k:k + width]) + "    /* row */" for k in range(0, len(flat), width))

    out = ["## RNAfold parameter file v2.0", "", "/* synthetic tables for tests: not a thermodynamic model */", ""]

    def section(name, body, with_dh=True):
        out.extend(["# " + name, "/* header comment", "   over two lines */", body, ""])
        if with_dh:   # the enthalpies must be ignored by a 37 C reader: fill them with poison
            out.extend(["# " + name + "_enthalpies", body.replace("-", "").replace("INF", "7777"), ""])

    section("stack", rows(T["stack"][1:, 1:], 7))
    for sec, name in MISMATCH_SECTIONS.items():
        section(sec, rows(T[name][1:], 5))
    section("dangle5", rows(T["dangle5"][1:], 5))
    section("dangle3", rows(T["dangle3"][1:], 5))
    section("int11", rows(T["int11"][1:, 1:], 5))
    section("int21", rows(T["int21"][1:, 1:], 5))
    section("int22", rows(T["int22"][1:7, 1:7, 1:, 1:, 1:, 1:], 4))
    section("hairpin", rows(T["hairpin"], 10))
    section("bulge", rows(T["bulge"], 10))
    section("interior", rows(T["interior"], 10))
    section("NINIO", "/* Ninio = MIN(max, m*|n1-n2| */\n/*\t    m\t  m_dH     max  */\n\t   %d\t   320\t   %d" % (T["ninio"], T["max_ninio"]), False)
    section("ML_params", "/* cu cu_dH cc cc_dH ci ci_dH */\n   %d   0   %d   3000   %d   -220" % (T["ML_base"], T["ML_closing"], T["ML_intern"]), False)
    section("Misc", "/* DuplexInit DuplexInit_dH TerminalAU TerminalAU_dH lxc lxc_dH */\n   %d   360   %d   370   %f   0.000000" % (T["DuplexInit"], T["TerminalAU"], T["lxc"]), False)
    for name in ("Hexaloops", "Tetraloops", "Triloops"):
        out.append("# " + name)
        out.extend("%s   %d   %d" % (k, v, 2 * v + 100) for k, v in T[name].items())
        out.append("")
    out.append("#END")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def read_par(path, into=None):
    """ViennaRNA parameter file -> tables (values of sections that are absent, and DEF entries, keep what `into` holds)"""
    T = into if into is not None else empty_tables()
    text = open(path).read()
    while "/*" in text:
        a = text.index("/*")
        b = text.find("*/", a + 2)
        text = text[:a] + (text[b + 2:] if b >= 0 else "")
    if "v2.0" in text.split("\n", 1)[0]:
        T["v20"] = True
    sections, name = {}, None
    for line in text.splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith("##"):
            continue
        if line.startswith("#"):
            name = line[1:].split()[0] if line[1:].split() else None
            sections.setdefault(name, [])
            continue
        if name:
            sections[name].append(line.split())

    def put(arr, index_iter, lines):
        toks = [t for ln in lines for t in ln]
        idx = list(index_iter)
        if len(toks) != len(idx):
            raise ValueError("section with %d values, expected %d" % (len(toks), len(idx)))
        for t, ix in zip(toks, idx):
            if t != "DEF":
                arr[ix] = INF if t == "INF" else int(t)

    def have(*names):
        return next((n for n in names if n in sections), None)

    n = have("stack", "stack_energies")
    if n:
        t0 = 0 if sum(map(len, sections[n])) == 64 else 1
        put(T["stack"], itertools.product(range(t0, 8), range(t0, 8)), sections[n])
    for sec, key in MISMATCH_SECTIONS.items():
        if sec in sections:
            t0 = 0 if sum(map(len, sections[sec])) == 200 else 1
            put(T[key], itertools.product(range(t0, 8), range(5), range(5)), sections[sec])
    for key in ("dangle5", "dangle3"):
        if key in sections:
            t0 = 0 if sum(map(len, sections[key])) == 40 else 1
            put(T[key], itertools.product(range(t0, 8), range(5)), sections[key])
    n = have("int11", "int11_energies")
    if n:
        put(T["int11"], itertools.product(range(1, 8), range(1, 8), range(5), range(5)), sections[n])
    n = have("int21", "int21_energies")
    if n:
        put(T["int21"], itertools.product(range(1, 8), range(1, 8), range(5), range(5), range(5)), sections[n])
    n = have("int22", "int22_energies")
    if n:
        tmax = 8 if sum(map(len, sections[n])) == 12544 else 7
        put(T["int22"], itertools.product(range(1, tmax), range(1, tmax), range(1, 5), range(1, 5), range(1, 5), range(1, 5)), sections[n])
    for sec, key in (("hairpin", "hairpin"), ("bulge", "bulge"), ("interior", "interior"), ("internal_loop", "interior")):
        if sec in sections:
            put(T[key], ((u,) for u in range(31)), sections[sec])
    if "NINIO" in sections:
        v = [t for ln in sections["NINIO"] for t in ln]
        T["ninio"], T["max_ninio"] = int(v[0]), int(v[2] if len(v) == 3 else v[1])
    if "ML_params" in sections:
        v = [t for ln in sections["ML_params"] for t in ln]
        if len(v) == 6:
            T["ML_base"], T["ML_closing"], T["ML_intern"] = int(v[0]), int(v[2]), int(v[4])
        else:
            T["ML_base"], T["ML_closing"], T["ML_intern"], T["TerminalAU"] = (int(x) for x in v[:4])
    if "Misc" in sections:
        v = [t for ln in sections["Misc"] for t in ln]
        T["DuplexInit"], T["TerminalAU"] = int(v[0]), int(v[2])
        if len(v) >= 5:
            T["lxc"] = float(v[4])
    for name in ("Tetraloops", "Triloops", "Hexaloops"):
        if name in sections:
            T[name] = {ln[0]: (INF if ln[1] == "INF" else int(ln[1])) for ln in sections[name] if len(ln) >= 2}
    return T


# ---------------------------------------------------------------- energies (10 cal/mol integers), ViennaRNA 2.x
def scaled(T):
    """get_scaled_parameters at 37 C: the *37 tables as they are, dangles and mismatchM / mismatchExt clipped to <= 0"""
    P = dict(T)
    for k in ("dangle5", "dangle3", "mismatchM", "mismatchExt"):
        P[k] = np.minimum(T[k], 0)
    return P


def E_ExtLoop(P, t, si1, sj1):
    e = 0
    if si1 >= 0 and sj1 >= 0:
        e += P["mismatchExt"][t, si1, sj1]
    elif si1 >= 0:
        e += P["dangle5"][t, si1]
    elif sj1 >= 0:
        e += P["dangle3"][t, sj1]
    if t > 2:
        e += P["TerminalAU"]
    return int(e)


def E_IntLoop(P, n1, n2, t, t2, si1, sj1, sp1, sq1):
    nl, ns = max(n1, n2), min(n1, n2)
    if nl == 0:
        return int(P["stack"][t, t2])
    if ns == 0:
        e = P["bulge"][nl] if nl <= MAXLOOP else P["bulge"][30] + int(P["lxc"] * math.log(nl / 30.0))
        if nl == 1:
            e += P["stack"][t, t2]
        else:
            e += (P["TerminalAU"] if t > 2 else 0) + (P["TerminalAU"] if t2 > 2 else 0)
        return int(e)
    if ns == 1:
        if nl == 1:
            return int(P["int11"][t, t2, si1, sj1])
        if nl == 2:
            return int(P["int21"][t, t2, si1, sq1, sj1] if n1 == 1 else P["int21"][t2, t, sq1, si1, sp1])
        e = P["interior"][nl + 1] if nl + 1 <= MAXLOOP else P["interior"][30] + int(P["lxc"] * math.log((nl + 1) / 30.0))
        e += min(P["max_ninio"], (nl - ns) * P["ninio"])
        return int(e + P["mismatch1nI"][t, si1, sj1] + P["mismatch1nI"][t2, sq1, sp1])
    if ns == 2:
        if nl == 2:
            return int(P["int22"][t, t2, si1, sp1, sq1, sj1])
        if nl == 3:
            return int(P["interior"][5] + P["ninio"] + P["mismatch23I"][t, si1, sj1] + P["mismatch23I"][t2, sq1, sp1])
    u = nl + ns
    e = P["interior"][u] if u <= MAXLOOP else P["interior"][30] + int(P["lxc"] * math.log(u / 30.0))
    e += min(P["max_ninio"], (nl - ns) * P["ninio"])
    return int(e + P["mismatchI"][t, si1, sj1] + P["mismatchI"][t2, sq1, sp1])


def logadd(x, y):
    if x == -math.inf:
        return y
    if y == -math.inf:
        return x
    return x + math.log1p(math.exp(y - x)) if x > y else y + math.log1p(math.exp(x - y))


def pf_duplex(T, seq1, seq2):
    """pf_duplex.c:42-206 (HAVE_VIENNA20 branch): returns (Esum_fw, Esum_bk, pr[(n1+1),(n2+1)])"""
    P = scaled(T)
    S1, S2 = encode(seq1), encode(seq2)
    n1, n2 = len(seq1), len(seq2)
    w = lambda E: -E * 10.0 / KT
    fw = np.full((n1 + 1, n2 + 1), -math.inf)
    bk = np.full((n1 + 1, n2 + 1), -math.inf)
    esum = -math.inf
    for i in range(1, n1 + 1):                       # pf_duplex_fw, :128-166
        for j in range(n2, 0, -1):
            t = PAIR[S1[i], S2[j]]
            if not t:
                continue
            E = P["DuplexInit"] + E_ExtLoop(P, t, S1[i - 1] if i > 1 else -1, S2[j + 1] if j < n2 else -1)
            v = w(E)
            k = i - 1
            while k > 0 and k > i - MAXLOOP - 2:
                for l in range(j + 1, n2 + 1):
                    if i - k + l - j - 2 > MAXLOOP:
                        break
                    t2 = PAIR[S1[k], S2[l]]
                    if t2 and fw[k, l] > -math.inf:
                        v = logadd(v, fw[k, l] + w(E_IntLoop(P, i - k - 1, l - j - 1, t2, RTYPE[t], S1[k + 1], S2[l - 1], S1[i - 1], S2[j + 1])))
                k -= 1
            fw[i, j] = v
            esum = logadd(esum, v + w(E_ExtLoop(P, RTYPE[t], S2[j - 1] if j > 1 else -1, S1[i + 1] if i < n1 else -1)))
    esum_bk = -math.inf
    for i in range(n1, 0, -1):                       # pf_duplex_bk, :168-206 (push form)
        for j in range(1, n2 + 1):
            t = PAIR[S1[i], S2[j]]
            if not t:
                continue
            bk[i, j] = logadd(bk[i, j], w(E_ExtLoop(P, RTYPE[t], S2[j - 1] if j > 1 else -1, S1[i + 1] if i < n1 else -1)))
            k = i - 1
            while k > 0 and k > i - MAXLOOP - 2:
                for l in range(j + 1, n2 + 1):
                    if i - k + l - j - 2 > MAXLOOP:
                        break
                    t2 = PAIR[S1[k], S2[l]]
                    if t2:
                        bk[k, l] = logadd(bk[k, l], bk[i, j] + w(E_IntLoop(P, i - k - 1, l - j - 1, t2, RTYPE[t], S1[k + 1], S2[l - 1], S1[i - 1], S2[j + 1])))
                k -= 1
            E = P["DuplexInit"] + E_ExtLoop(P, t, S1[i - 1] if i > 1 else -1, S2[j + 1] if j < n2 else -1)
            esum_bk = logadd(esum_bk, bk[i, j] + w(E))
    pr = np.zeros((n1 + 1, n2 + 1))
    for i in range(1, n1 + 1):
        for j in range(1, n2 + 1):
            if PAIR[S1[i], S2[j]] and fw[i, j] > -math.inf and bk[i, j] > -math.inf:
                pr[i, j] = math.exp(fw[i, j] + bk[i, j] - esum)
    return esum, esum_bk, pr


def brute_duplex(T, seq1, seq2):
    """every chain of inter-molecular pairs (i ascending in s1, j descending in s2, loops <= MAXLOOP): (log Z, pr)"""
    P = scaled(T)
    S1, S2 = encode(seq1), encode(seq2)
    n1, n2 = len(seq1), len(seq2)
    cells = [(i, j) for i in range(1, n1 + 1) for j in range(1, n2 + 1) if PAIR[S1[i], S2[j]]]
    Z = 0.0
    pr = np.zeros((n1 + 1, n2 + 1))

    def extend(chain, energy):
        nonlocal Z
        i, j = chain[-1]
        t = PAIR[S1[i], S2[j]]
        e_all = energy + E_ExtLoop(P, RTYPE[t], S2[j - 1] if j > 1 else -1, S1[i + 1] if i < n1 else -1)
        wgt = math.exp(-e_all * 10.0 / KT)
        Z += wgt
        for c in chain:
            pr[c] += wgt
        for (p, q) in cells:          # next pair downstream: p > i, q < j
            if p > i and q < j and (p - i - 1) + (j - q - 1) <= MAXLOOP:
                tp = PAIR[S1[p], S2[q]]
                e = E_IntLoop(P, p - i - 1, j - q - 1, t, RTYPE[tp], S1[i + 1], S2[j - 1], S1[p - 1], S2[q + 1])
                extend(chain + [(p, q)], energy + e)

    for (i, j) in cells:
        t = PAIR[S1[i], S2[j]]
        extend([(i, j)], P["DuplexInit"] + E_ExtLoop(P, t, S1[i - 1] if i > 1 else -1, S2[j + 1] if j < n2 else -1))
    return (math.log(Z) if Z > 0 else -math.inf), (pr / Z if Z > 0 else pr)


# ---------------------------------------------------------------- pf_fold under 2.x energies (dangles = 2), by enumeration
def smooth_w(E):
    """log Boltzmann weight of a dangle / mismatchM / mismatchExt energy: exp(SMOOTH(-E)*10/kT), params.c"""
    X = -float(E)
    x = X / 10.0
    if x < -1.2283697:
        s = 0.0
    elif x > 0.8660254:
        s = X
    else:
        s = 10.0 * 0.38490018 * (math.sin(x - 0.34242663) + 1.0) ** 2
    return s * 10.0 / KT


def w_stem(T, which, t, si1, sj1):
    """exp_E_ExtLoop (which = 'mismatchExt') / the mismatch part of exp_E_MLstem ('mismatchM'): log weight, TerminalAU included;
    si1 / sj1 = 5' / 3' neighbour letter, <= 0 when there is none (unknown letters count as none, as in the product)"""
    v = 0.0
    if si1 > 0 and sj1 > 0:
        v = smooth_w(T[which][t, si1, sj1])
    elif si1 > 0:
        v = smooth_w(T["dangle5"][t, si1])
    elif sj1 > 0:
        v = smooth_w(T["dangle3"][t, sj1])
    return v - (T["TerminalAU"] * 10.0 / KT if t > 2 else 0.0)


def w_hairpin(T, seq, S, i, j):
    """exp_E_Hairpin: letters i..j (1-based), closed by (i, j)"""
    u = j - i - 1
    t = PAIR[S[i], S[j]]
    e = T["hairpin"][u] if u <= 30 else T["hairpin"][30] + int(T["lxc"] * math.log(u / 30.0))
    sub = seq[i - 1:j].upper().replace("T", "U")
    if u == 3:
        if sub in T["Triloops"]:
            return -T["Triloops"][sub] * 10.0 / KT
        return -(e + (T["TerminalAU"] if t > 2 else 0)) * 10.0 / KT
    if u == 4 and sub in T["Tetraloops"]:
        return -T["Tetraloops"][sub] * 10.0 / KT
    if u == 6 and sub in T["Hexaloops"]:
        return -T["Hexaloops"][sub] * 10.0 / KT
    return -(e + T["mismatchH"][t, S[i + 1], S[j - 1]]) * 10.0 / KT


def structures(n, S, min_hairpin=3):
    """all secondary structures over letters 1..n as sorted tuples of (i, j), canonical pair types only"""
    def rec(lo, hi):
        if hi - lo < min_hairpin + 1:
            return [()]
        out = list(rec(lo + 1, hi))                      # lo unpaired
        for k in range(lo + min_hairpin + 1, hi + 1):   # lo pairs with k
            if PAIR[S[lo], S[k]]:
                for a in rec(lo + 1, k - 1):
                    for b in rec(k + 1, hi):
                        out.append(((lo, k),) + a + b)
        return out
    return rec(1, n)


def brute_fold(T, seq, max_w=0):
    """pf_fold by enumeration under the 2.x loop energies with dangles = 2: (log Z, bp dict[, up]) with
    up[i][w] = P(letters i+1 .. i+1+w unpaired), i and w from 0 (the layout of rh_fold), when max_w > 0"""
    S = encode(seq)
    n = len(seq)
    P = T
    w = lambda E: -E * 10.0 / KT
    Z, bp = 0.0, {}
    up = np.zeros((n, max(max_w, 1)))
    for st in structures(n, S):
        partner = {}
        for (i, j) in st:
            partner[i] = j
        logw = 0.0

        def children(lo, hi):
            c, k = [], lo
            while k <= hi:
                if k in partner:
                    c.append((k, partner[k]))
                    k = partner[k] + 1
                else:
                    k += 1
            return c

        for (i, j) in children(1, n):      # exterior loop
            t = PAIR[S[i], S[j]]
            logw += w_stem(T, "mismatchExt", t, S[i - 1] if i > 1 else -1, S[j + 1] if j < n else -1)
        ok = True
        for (i, j) in st:
            t = PAIR[S[i], S[j]]
            ch = children(i + 1, j - 1)
            if not ch:
                logw += w_hairpin(T, seq, S, i, j)
            elif len(ch) == 1:
                p, q = ch[0]
                if (p - i - 1) + (j - q - 1) > MAXLOOP:
                    ok = False
                    break
                logw += w(E_IntLoop(P, p - i - 1, j - q - 1, t, RTYPE[PAIR[S[p], S[q]]], S[i + 1], S[j - 1], S[p - 1], S[q + 1]))
            else:
                unpaired = (j - i - 1) - sum(q - p + 1 for p, q in ch)
                logw += w(T["ML_closing"] + T["ML_intern"]) + w_stem(T, "mismatchM", RTYPE[t], S[j - 1], S[i + 1]) + unpaired * w(T["ML_base"])
                for (p, q) in ch:
                    logw += w(T["ML_intern"]) + w_stem(T, "mismatchM", PAIR[S[p], S[q]], S[p - 1], S[q + 1])
        if not ok:
            continue
        wgt = math.exp(logw)
        Z += wgt
        for c in st:
            bp[c] = bp.get(c, 0.0) + wgt
        if max_w > 0:
            paired = set(partner) | set(partner.values())
            for i in range(n):
                for ww in range(max_w):
                    if i + ww >= n or (i + 1 + ww) in paired:
                        break
                    up[i, ww] += wgt
    if max_w > 0:
        return math.log(Z), {k: v / Z for k, v in bp.items()}, up / Z
    return math.log(Z), {k: v / Z for k, v in bp.items()}


def brute_cofold(T, seq1, seq2):
    """co_pf_fold by enumeration over s1+s2 (cut after letter n1), 2.x energies, dangles = 2: (log Z, hp[(n1+1),(n2+1)]).
    The loop whose backbone holds the missing gap is exterior-like: every stem in it (the closing pair seen from inside
    included) scores as a stem of the exterior loop, a neighbour letter counts only if it sits on the same strand, and
    there is no loop-size rule for it; pairs still need 3 letters between them (TURN applies across the gap too)."""
    n1, n2 = len(seq1), len(seq2)
    seq = seq1 + seq2
    S = encode(seq)
    n, cut = n1 + n2, n1
    w = lambda E: -E * 10.0 / KT
    nb5 = lambda p: S[p - 1] if (p - 1 >= 1 and p - 1 != cut) else -1      # letter before p, same strand
    nb3 = lambda q: S[q + 1] if (q + 1 <= n and q != cut) else -1          # letter after q, same strand
    Z = 0.0
    hp = np.zeros((n1 + 1, n2 + 1))
    for st in structures(n, S):
        partner = {i: j for (i, j) in st}

        def children(lo, hi):
            c, k = [], lo
            while k <= hi:
                if k in partner:
                    c.append((k, partner[k]))
                    k = partner[k] + 1
                else:
                    k += 1
            return c

        logw = 0.0
        for (p, q) in children(1, n):
            logw += w_stem(T, "mismatchExt", PAIR[S[p], S[q]], nb5(p), nb3(q))
        ok = True
        for (i, j) in st:
            t = PAIR[S[i], S[j]]
            ch = children(i + 1, j - 1)
            nicked = i <= cut < j and not any(p <= cut < q for p, q in ch)
            if nicked:
                logw += w_stem(T, "mismatchExt", RTYPE[t], nb5(j), nb3(i))
                for (p, q) in ch:
                    logw += w_stem(T, "mismatchExt", PAIR[S[p], S[q]], nb5(p), nb3(q))
            elif not ch:
                logw += w_hairpin(T, seq, S, i, j)
            elif len(ch) == 1:
                p, q = ch[0]
                if (p - i - 1) + (j - q - 1) > MAXLOOP:
                    ok = False
                    break
                logw += w(E_IntLoop(T, p - i - 1, j - q - 1, t, RTYPE[PAIR[S[p], S[q]]], S[i + 1], S[j - 1], S[p - 1], S[q + 1]))
            else:
                unpaired = (j - i - 1) - sum(q - p + 1 for p, q in ch)
                logw += w(T["ML_closing"] + T["ML_intern"]) + w_stem(T, "mismatchM", RTYPE[t], S[j - 1], S[i + 1]) + unpaired * w(T["ML_base"])
                for (p, q) in ch:
                    logw += w(T["ML_intern"]) + w_stem(T, "mismatchM", PAIR[S[p], S[q]], S[p - 1], S[q + 1])
        if not ok:
            continue
        wgt = math.exp(logw)
        Z += wgt
        for (i, j) in st:
            if i <= cut < j:
                hp[i, j - cut] += wgt
    return math.log(Z), hp / Z
