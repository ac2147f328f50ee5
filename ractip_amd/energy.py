"""Free energy of a given structure under the Vienna-BL model of this package (kcal/mol).

What RactIP obtains from ViennaRNA's energy_of_structure(): e1/e2 of the two molecules' own structures
(`/root/reference/src/ractip.cpp:1254, 1299, 1457`) and e3 of the interaction (`energy_of_duplex`, `:1529-1559`: both
sequences concatenated with a cut point, internal pairs erased, '[' ']' turned into '(' ')').  Caller-side code of SURVEY
8f-4; the third-party evaluator is absent and unversioned, so -- like the probability layer of this model -- this is a
PARITY-UNPINNED restatement: the energy of a structure here is exactly the -kT*log of the Boltzmann weight that
ractip_amd's partition functions (mccaskill_vlin.hip / mccaskill_vienna.hip) give it, loop by loop: the BL* tables of
src/boltzmann_param.c, ViennaRNA-1.8 loop energies, dangles on both sides of every multi/exterior stem (smoothed as in
part_func.c), TerminalAU, tetraloop bonuses; with a cut point the loop that holds the missing backbone gap is exterior-like
and no dangle crosses it.  tests/test_energy.py checks that sum over all structures of exp(-E/kT) is the partition function.
"""
import math
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
_PT = np.zeros((5, 5), dtype=int)
for (_a, _b), _t in {(2, 3): 1, (3, 2): 2, (3, 4): 3, (4, 3): 4, (1, 4): 5, (4, 1): 6}.items():
    _PT[_a, _b] = _t
_RT = [0, 2, 1, 4, 3, 6, 5, 7]
_CODE = {"A": 1, "C": 2, "G": 3, "U": 4, "T": 4}
KT = (37.0 + 273.15) * 1.98717   # cal/mol (src/ractip.cpp:262, src/pf_duplex.c:73)
MAXLOOP = 30


def _smooth(x):   # SMOOTH of part_func.c, x in 10 cal/mol
    y = x / 10.0
    if y < -1.2283697:
        return 0.0
    if y > 0.8660254:
        return x
    t = math.sin(y - 0.34242663) + 1.0
    return 10.0 * 0.38490018 * t * t


class Model:
    def __init__(self, path=None):
        path = path or os.path.join(PKG, "data", "vienna_bl_star.params")
        tabs, tetra = {}, {}
        toks = [l for l in open(path) if l.strip() and not l.startswith("#")]
        k = 0
        while k < len(toks):
            name, cnt = toks[k].split()[:2]
            cnt = int(cnt)
            k += 1
            if name == "tetraloops":
                for line in toks[k:k + cnt]:
                    sq, e = line.split()
                    tetra[sq] = int(e)
                k += cnt
                continue
            vals = []
            while len(vals) < cnt:
                vals += [int(v) for v in toks[k].split()]
                k += 1
            tabs[name] = vals
        g = lambda name, shape: np.array(tabs[name]).reshape(shape)
        self.stack = np.zeros((8, 8), int); self.stack[1:, 1:] = g("stack37", (7, 7))
        self.mmI = np.zeros((8, 5, 5), int); self.mmI[1:] = g("mismatchI37", (7, 5, 5))
        self.mmH = np.zeros((8, 5, 5), int); self.mmH[1:] = g("mismatchH37", (7, 5, 5))
        d5, d3 = g("dangle5_37", (8, 5)), g("dangle3_37", (8, 5))
        self.int11 = np.zeros((8, 8, 5, 5), int); self.int11[1:, 1:] = g("int11_37", (7, 7, 5, 5))
        self.int21 = np.zeros((8, 8, 5, 5, 5), int); self.int21[1:, 1:] = g("int21_37", (7, 7, 5, 5, 5))
        self.int22 = np.zeros((8, 8, 5, 5, 5, 5), int); self.int22[1:, 1:, 1:, 1:, 1:, 1:] = g("int22_37", (7, 7, 4, 4, 4, 4))
        self.hairpin, self.bulge, self.internal = tabs["hairpin37"], tabs["bulge37"], tabs["internal_loop37"]
        self.ml_base, self.ml_closing, self.ml_intern, self.tau = tabs["MLparams"]
        self.ninio, self.max_ninio = tabs["ninio"]
        self.tetra = tetra
        self.lxc = 107.856
        # stem dangles: smoothed, TerminalAU folded into the 3' side, letter code 0 = no neighbour
        self.d5x = np.zeros((8, 5)); self.d3x = np.zeros((8, 5))
        for t in range(8):
            for x in range(5):
                self.d5x[t, x] = -_smooth(-float(d5[t, x])) if x else 0.0
                self.d3x[t, x] = (-_smooth(-float(d3[t, x])) if x else 0.0) + (self.tau if t > 2 else 0)

    def loop(self, n1, n2, t, t2, si1, sj1, sp1, sq1):
        nl, ns = max(n1, n2), min(n1, n2)
        if nl == 0:
            return self.stack[t, t2]
        if ns == 0:
            e = self.bulge[nl]
            return e + self.stack[t, t2] if nl == 1 else e + (self.tau if t > 2 else 0) + (self.tau if t2 > 2 else 0)
        if ns == 1 and nl == 1:
            return self.int11[t, t2, si1, sj1]
        if ns == 1 and nl == 2:
            return self.int21[t, t2, si1, sq1, sj1] if n1 == 1 else self.int21[t2, t, sq1, si1, sp1]
        if n1 == 2 and n2 == 2:
            return self.int22[t, t2, si1, sp1, sq1, sj1]
        return self.internal[n1 + n2] + min(self.max_ninio, (nl - ns) * self.ninio) + self.mmI[t, si1, sj1] + self.mmI[t2, sq1, sp1]


_default = None


def pair_table(structure):
    pt, st = [0] * (len(structure) + 1), []
    for i, ch in enumerate(structure, 1):
        if ch == "(":
            st.append(i)
        elif ch == ")":
            j = st.pop()
            pt[i], pt[j] = j, i
    if st:
        raise ValueError("unbalanced structure")
    return pt


def energy_of_structure(seq, structure, cut=0, model=None, noncanonical=False):
    """kcal/mol; cut = n1 > 0: seq = s1+s2, the backbone gap after letter n1 does not exist (energy_of_duplex's cut_point-1).
    Returns +inf for a structure outside the ensemble (unpairable letters, hairpin < 3, interior loop > 30).
    noncanonical=True: a pair of letters that cannot pair is scored as pair type 7, as ViennaRNA's evaluator does (the
    single-sequence programme of the z-score loop has no crossing constraint, so its bracket string can match such letters)."""
    global _default
    M = model or _default or Model()
    if model is None:
        _default = M
    n = len(seq)
    S = [0] + [_CODE.get(c.upper(), 0) for c in seq] + [0, 0]
    pt = pair_table(structure)
    ok = lambda g: cut == 0 or g != cut   # letters g, g+1 are backbone neighbours
    PT = lambda a, b: (_PT[a, b] or 7) if noncanonical else _PT[a, b]
    stem = lambda p, q: M.d5x[PT(S[p], S[q]), S[p - 1] if ok(p - 1) else 0] + M.d3x[PT(S[p], S[q]), S[q + 1] if ok(q) else 0]
    E = 0.0

    def loop(a, b):
        nonlocal E
        t = PT(S[a], S[b])
        if not t or b - a < 4:
            E = math.inf
            return
        stems, unpaired, nick, x = [], 0, False, a + 1
        while x < b:
            if cut and x - 1 == cut:
                nick = True
            if pt[x] > x:
                stems.append((x, pt[x])); x = pt[x] + 1
            else:
                unpaired += 1; x += 1
        if cut and b - 1 == cut:
            nick = True
        tt = _RT[t]
        if nick:       # exterior-like: only the stems' own terms, the closing pair seen from inside
            E += M.d3x[tt, S[a + 1] if ok(a) else 0] + M.d5x[tt, S[b - 1] if ok(b - 1) else 0]
            E += sum(stem(p, q) for p, q in stems)
        elif not stems:
            u = b - a - 1
            e = M.hairpin[u] if u <= 30 else M.hairpin[30] + M.lxc * math.log(u / 30.0)
            if u == 4:
                e += M.tetra.get("".join("_ACGU"[S[k]] for k in range(a, a + 6)), 0)
            E += e + ((M.tau if t > 2 else 0) if u == 3 else M.mmH[t, S[a + 1], S[b - 1]])
        elif len(stems) == 1:
            p, q = stems[0]
            if unpaired > MAXLOOP:
                E = math.inf
                return
            E += M.loop(p - a - 1, b - q - 1, t, _RT[PT(S[p], S[q])], S[a + 1], S[b - 1], S[p - 1], S[q + 1])
        else:
            E += M.ml_closing + M.ml_intern + M.d3x[tt, S[a + 1]] + M.d5x[tt, S[b - 1]] + M.ml_base * unpaired
            E += sum(M.ml_intern + stem(p, q) for p, q in stems)
        for p, q in stems:
            loop(p, q)

    x = 1
    while x <= n:
        if pt[x] > x:
            if not PT(S[x], S[pt[x]]):
                return math.inf
            E += stem(x, pt[x])
            loop(x, pt[x])
            x = pt[x] + 1
        else:
            x += 1
    return E / 100.0


def energy_of_duplex(s1, s2, r1, r2, model=None, noncanonical=True):
    """RactIP::energy_of_duplex (src/ractip.cpp:1529-1559): internal pairs erased, '[' ']' become the pairs of s1+s2."""
    rr = "".join({"(": ".", ")": ".", "[": "(", "]": ")"}.get(ch, ch) for ch in r1 + r2)
    return energy_of_structure(s1 + s2, rr, cut=len(s1), model=model, noncanonical=noncanonical)
