"""Joint-structure stage above the probability matrices: RactIP::solve's 0/1 programme and its decoding.

The caller side of the hot path.  `/root/reference/src/ractip.cpp:551-1353` thresholds bp / hp / up,
builds an integer programme through the thin solver interface `class IP` (`src/ip.h:25-44`) and decodes the optimum into
two bracket strings.  This module restates that model over the matrices this package computes, so that a pair can be
taken from sequences to the joint structure without the reference binary (which needs ViennaRNA and a MIP library that
are not installed).  `IPModel` keeps `class IP`'s interface (make_variable / make_constraint / add_constraint / solve /
get_value, bound kinds FR LO UP DB FX); its backend is HiGHS -- one of the five backends the reference itself supports
(`src/ip.cpp:490-622`) -- as bundled with SciPy (`scipy.optimize.milp`).

Weights are formed in float exactly where the reference does (`VF` containers and `float` thresholds,
`src/ractip.cpp:82-83, 157-161`), so two probability sources that agree after narrowing to float give the same
programme and therefore the same structure.
"""
import numpy as np

FR, LO, UP, DB, FX = range(5)


class IPModel:
    """`class IP` of src/ip.h:25-44 (maximisation, binary columns) on SciPy's HiGHS."""

    def __init__(self):
        self.obj = []
        self.rows = []       # (lo, hi)
        self.entries = []    # (row, col, val)

    def make_variable(self, coef):
        self.obj.append(float(coef))
        return len(self.obj) - 1

    def make_constraint(self, bnd, l, u):
        lo, hi = {FR: (-np.inf, np.inf), LO: (l, np.inf), UP: (-np.inf, u), DB: (l, u), FX: (l, l)}[bnd]
        self.rows.append((lo, hi))
        return len(self.rows) - 1

    def add_constraint(self, row, col, val):
        self.entries.append((row, col, float(val)))

    def solve(self):
        from scipy.optimize import Bounds, LinearConstraint, milp
        from scipy.sparse import coo_matrix
        n, m = len(self.obj), len(self.rows)
        self.x = np.zeros(n)
        if n == 0:
            return 0.0
        cons = []
        if m and self.entries:
            r, c, v = zip(*self.entries)
            A = coo_matrix((v, (r, c)), shape=(m, n)).tocsr()   # duplicate (row, col) entries add up, as in the solvers' APIs
            lo = np.array([b[0] for b in self.rows])
            hi = np.array([b[1] for b in self.rows])
            cons = [LinearConstraint(A, lo, hi)]
        res = milp(c=-np.asarray(self.obj), constraints=cons, integrality=np.ones(n), bounds=Bounds(0, 1),
                   options={"mip_rel_gap": 0.0})
        if res.x is None:
            raise RuntimeError("integer programme has no solution: %s" % res.message)
        self.x = res.x
        return float(-res.fun)

    def get_value(self, col):
        return self.x[col]


class Options:
    """Defaults of src/cmdline.c:151-186 as mapped onto members at src/ractip.cpp:1474-1498."""

    def __init__(self, **kw):
        self.alpha, self.beta = 0.7, 0.0
        self.th_ss, self.th_hy, self.th_ac = 0.5, 0.1, 0.003
        self.acc_max, self.acc_num = False, 1
        self.max_w, self.min_w = 15, 5
        self.in_pk = True                 # --no-pk off
        self.stacking_constraints = True  # --allow-isolated off
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option %r" % k)
            setattr(self, k, v)


def _tri(n, i):
    return i * (2 * (n + 1) - i - 1) // 2


def solve(s1, s2, bp1, bp2, hp, up1=None, up2=None, opt=None):
    """RactIP::solve after the probability layer (src/ractip.cpp:551-1353, structure constraints off).

    bp1/bp2: triangular pair matrices in the reference layout; hp: (n1+1) x (n2+1), 1-based; up1/up2: n x max_w
    accessibility (or None: accessibility disabled, as with --min-w 0).  Returns (r1, r2, objective)."""
    opt = opt or Options()
    f32 = np.float32
    n1, n2 = len(s1), len(s2)
    bp1, bp2, hp = np.asarray(bp1, f32), np.asarray(bp2, f32), np.asarray(hp, f32)
    th_ss, th_hy, th_ac, alpha, beta = f32(opt.th_ss), f32(opt.th_hy), f32(opt.th_ac), f32(opt.alpha), f32(opt.beta)
    enable_acc = opt.min_w > 1 and opt.max_w >= opt.min_w and up1 is not None and up2 is not None   # :526
    structure = not opt.acc_max                                                                      # :527-528
    ip = IPModel()

    def internal(n, bp):   # :551-568 / :572-589
        x = -np.ones((n, n), dtype=np.int64)
        xx = [[] for _ in range(n)]
        x_un = [-1] * n
        if structure:
            for j in range(1, n):
                for i in range(j - 1, -1, -1):
                    p = bp[_tri(n, i + 1) + (j + 1)]
                    if p > th_ss:
                        x[i][j] = x[j][i] = ip.make_variable(f32(p - th_ss))
                        xx[i].append(j)
            x_un = [ip.make_variable(0.0) for _ in range(n)]
        return x, xx, x_un

    x, xx, x_un = internal(n1, bp1)
    y, yy, y_un = internal(n2, bp2)
    z = -np.ones((n1, n2), dtype=np.int64)   # :592-612
    zz = [[] for _ in range(n1)]
    for i in range(n1):
        for j in range(n2):
            p = hp[i + 1][j + 1]
            if p > th_hy:
                z[i][j] = ip.make_variable(f32(alpha * f32(p - th_hy)))
                zz[i].append(j)
    z_un1 = [ip.make_variable(0.0) for _ in range(n1)]
    z_un2 = [ip.make_variable(0.0) for _ in range(n2)]

    def regions(n, up):   # :614-655
        v, vv = [], []
        if enable_acc:
            up = np.asarray(up, f32).reshape(n, -1)
            for i in range(up.shape[0]):
                for j in range(opt.min_w - 1, up.shape[1]):
                    if up[i][j] > th_ac:
                        v.append(ip.make_variable(f32(beta * f32(up[i][j] - th_ac))))
                        vv.append((i, i + j))
        st = [ip.make_variable(0.0) for _ in range(n)]
        en = [ip.make_variable(0.0) for _ in range(n)]
        return v, vv, st, en

    v, vv, v_st, v_en = regions(n1, up1)
    w, ww, w_st, w_en = regions(n2, up2)

    def row_of(bnd, l, u, terms):
        r = ip.make_constraint(bnd, l, u)
        for col, val in terms:
            ip.add_constraint(r, col, val)
        return r

    # helper variables (:717-765): every letter is paired exactly once or marked unpaired, per kind of pair
    if structure:
        for i in range(n1):
            row_of(FX, 1, 1, [(x_un[i], 1)] + [(x[i][j], 1) for j in range(n1) if x[i][j] >= 0])
    for i in range(n1):
        row_of(FX, 1, 1, [(z_un1[i], 1)] + [(z[i][j], 1) for j in range(n2) if z[i][j] >= 0])
    if structure:
        for i in range(n2):
            row_of(FX, 1, 1, [(y_un[i], 1)] + [(y[i][j], 1) for j in range(n2) if y[i][j] >= 0])
    for i in range(n2):
        row_of(FX, 1, 1, [(z_un2[i], 1)] + [(z[j][i], 1) for j in range(n1) if z[j][i] >= 0])

    def region_links(n, v, vv, st, en):   # :767-801
        rs = [row_of(FX, 0, 0, [(st[i], -1)]) for i in range(n)]
        # rows are created start/end interleaved in the reference; the order of rows does not enter the optimum
        re_ = [row_of(FX, 0, 0, [(en[i], -1)]) for i in range(n)]
        for k, (a, b) in enumerate(vv):
            if b < n:
                ip.add_constraint(rs[a], v[k], 1)
                ip.add_constraint(re_[b], v[k], 1)

    if enable_acc:
        # a region that would run past the end of the sequence cannot be addressed (the reference indexes row_v_en[i+j]
        # out of range for it); pf_unstru leaves such entries at 0, so they never pass th_ac
        vv_ok = all(b < n1 for _, b in vv) and all(b < n2 for _, b in ww)
        if not vv_ok:
            raise ValueError("accessibility matrix has mass on regions past the end of a sequence")
        region_links(n1, v, vv, v_st, v_en)
        region_links(n2, w, ww, w_st, w_en)

    if not enable_acc:   # :804-830: a letter pairs at most once in total
        if structure:
            for i in range(n1):
                row_of(LO, 1, 0, [(x_un[i], 1), (z_un1[i], 1)])
            for i in range(n2):
                row_of(LO, 1, 0, [(y_un[i], 1), (z_un2[i], 1)])
    else:                # :831-985
        def cover(n, vlist, vvlist):
            c = [[] for _ in range(n)]
            for k, (a, b) in enumerate(vvlist):
                for i in range(a, b + 1):
                    c[i].append(vlist[k])
            return c
        c1, c2 = cover(n1, v, vv), cover(n2, w, ww)
        if structure:    # an internally paired letter is not inside an accessible region
            for i in range(n1):
                row_of(UP, 0, 0, [(x_un[i], -1)] + [(k, 1) for k in c1[i]])
        for i in range(n1):   # an externally paired letter is
            row_of(LO, 1, 0, [(z_un1[i], 1)] + [(k, 1) for k in c1[i]])
        if structure:
            for i in range(n2):
                row_of(UP, 0, 0, [(y_un[i], -1)] + [(k, 1) for k in c2[i]])
        for i in range(n2):
            row_of(LO, 1, 0, [(z_un2[i], 1)] + [(k, 1) for k in c2[i]])
        for i in range(n1):   # regions do not overlap ...
            row_of(UP, 0, 1, [(k, 1) for k in c1[i]])
        for i in range(1, n1):   # ... nor touch
            row_of(UP, 0, 1, [(v_en[i - 1], 1), (v_st[i], 1)])
        for i in range(n2):
            row_of(UP, 0, 1, [(k, 1) for k in c2[i]])
        for i in range(1, n2):
            row_of(UP, 0, 1, [(w_en[i - 1], 1), (w_st[i], 1)])
        if beta > 0.0:   # a rewarded region must hold an interaction (:937-958)
            for k, (a, b) in enumerate(vv):
                row_of(UP, 0, b - a + 1, [(v[k], 1)] + [(z_un1[i], 1) for i in range(a, b + 1)])
            for k, (a, b) in enumerate(ww):
                row_of(UP, 0, b - a + 1, [(w[k], 1)] + [(z_un2[i], 1) for i in range(a, b + 1)])
        if opt.acc_num > 0:   # :971-984 (stated twice in the reference, :987-996: the same rows again)
            row_of(UP, 0, opt.acc_num, [(k, 1) for k in v])
            row_of(UP, 0, opt.acc_num, [(k, 1) for k in w])

    # no crossing interactions (:999-1017)
    for i in range(n1):
        for k in range(i + 1, n1):
            for j in zz[i]:
                for l in zz[k]:
                    if j < l:
                        row_of(UP, 0, 1, [(z[i][j], 1), (z[k][l], 1)])

    def no_internal_pk(xm, xl):   # :1019-1062
        for i in range(len(xl)):
            for j in xl[i]:
                for k in range(i + 1, j):
                    for l in xl[k]:
                        if j < l:
                            row_of(UP, 0, 1, [(xm[i][j], 1), (xm[k][l], 1)])

    if opt.in_pk and structure:
        no_internal_pk(x, xx)
        no_internal_pk(y, yy)

    if opt.stacking_constraints:   # no isolated pairs (:1064-1177)
        def stack_rows(n, m):
            for i in range(n):   # pairs (j,i), j < i: needs (j', i-1) or (j', i+1)
                t = [(m[j][i], -1) for j in range(i) if m[j][i] >= 0]
                if i > 0:
                    t += [(m[j][i - 1], 1) for j in range(i - 1) if m[j][i - 1] >= 0]
                if i + 1 < n:
                    t += [(m[j][i + 1], 1) for j in range(i + 1) if m[j][i + 1] >= 0]
                row_of(LO, 0, 0, t)
            for i in range(n):   # pairs (i,j), j > i
                t = [(m[i][j], -1) for j in range(i + 1, n) if m[i][j] >= 0]
                if i > 0:
                    t += [(m[i - 1][j], 1) for j in range(i, n) if m[i - 1][j] >= 0]
                if i + 1 < n:
                    t += [(m[i + 1][j], 1) for j in range(i + 2, n) if m[i + 1][j] >= 0]
                row_of(LO, 0, 0, t)
        if structure:
            stack_rows(n1, x)
            stack_rows(n2, y)
        for i in range(n2):
            t = [(z[j][i], -1) for j in range(n1) if z[j][i] >= 0]
            if i > 0:
                t += [(z[j][i - 1], 1) for j in range(n1) if z[j][i - 1] >= 0]
            if i + 1 < n2:
                t += [(z[j][i + 1], 1) for j in range(n1) if z[j][i + 1] >= 0]
            row_of(LO, 0, 0, t)
        for i in range(n1):
            t = [(z[i][j], -1) for j in range(n2) if z[i][j] >= 0]
            if i > 0:
                t += [(z[i - 1][j], 1) for j in range(n2) if z[i - 1][j] >= 0]
            if i + 1 < n1:
                t += [(z[i + 1][j], 1) for j in range(n2) if z[i + 1][j] >= 0]
            row_of(LO, 0, 0, t)

    ea = ip.solve()

    # decode (:1227-1300)
    r1, r2 = ["."] * n1, ["."] * n2
    for i in range(n1):
        for j in range(n2):
            if z[i][j] >= 0 and ip.get_value(z[i][j]) > 0.5:
                r1[i], r2[j] = "[", "]"
    if structure and opt.in_pk:
        for r, m, n in ((r1, x, n1), (r2, y, n2)):
            for i in range(n):
                for j in range(i + 1, n):
                    if m[i][j] >= 0 and ip.get_value(m[i][j]) > 0.5:
                        assert r[i] == "." and r[j] == "."
                        r[i], r[j] = "(", ")"
    return "".join(r1), "".join(r2), ea


def solve_ss(s, bp, opt=None, usable=None):
    """RactIP::solve_ss (src/ractip.cpp:1355-1465): the single-sequence programme of the z-score loop -- pairs with
    bp > th_ss, each letter in at most one pair, no isolated pairs; NO crossing constraint (as in the reference, whose
    decoding then writes '(' ')' per selected pair).  Returns (structure, objective)."""
    opt = opt or Options()
    f32 = np.float32
    n = len(s)
    bp = np.asarray(bp, f32)
    th_ss = f32(opt.th_ss)
    u = usable if usable is not None else [True] * n
    ip = IPModel()
    x = -np.ones((n, n), dtype=np.int64)
    for j in range(1, n):
        if not u[j]:
            continue
        for i in range(j - 1, -1, -1):
            if not u[i]:
                continue
            p = bp[_tri(n, i + 1) + (j + 1)]
            if p > th_ss:
                x[i][j] = x[j][i] = ip.make_variable(f32(p - th_ss))
    for i in range(n):
        r = ip.make_constraint(UP, 0, 1)
        for j in range(n):
            if x[i][j] >= 0:
                ip.add_constraint(r, x[i][j], 1)
    if opt.stacking_constraints:
        for i in range(n):
            r = ip.make_constraint(LO, 0, 0)
            for j in range(i):
                if x[j][i] >= 0:
                    ip.add_constraint(r, x[j][i], -1)
            if i > 0:
                for j in range(i - 1):
                    if x[j][i - 1] >= 0:
                        ip.add_constraint(r, x[j][i - 1], 1)
            if i + 1 < n:
                for j in range(i + 1):
                    if x[j][i + 1] >= 0:
                        ip.add_constraint(r, x[j][i + 1], 1)
        for i in range(n):
            r = ip.make_constraint(LO, 0, 0)
            for j in range(i + 1, n):
                if x[i][j] >= 0:
                    ip.add_constraint(r, x[i][j], -1)
            if i > 0:
                for j in range(i, n):
                    if x[i - 1][j] >= 0:
                        ip.add_constraint(r, x[i - 1][j], 1)
            if i + 1 < n:
                for j in range(i + 2, n):
                    if x[i + 1][j] >= 0:
                        ip.add_constraint(r, x[i + 1][j], 1)
    ea = ip.solve()
    r = ["."] * n
    for i in range(n):
        for j in range(i + 1, n):
            if x[i][j] >= 0 and ip.get_value(x[i][j]) > 0.5:
                r[i], r[j] = "(", ")"
    return "".join(r), ea

