"""CPU suite: ViennaRNA parameter files and the two loop-energy semantics of the Vienna model (PARITY UNPINNED -- ViennaRNA is
absent, see oracle/vienna2x.py).  What is pinned here, without a GPU:
  * the product's loader (ractip_amd/csrc/vienna_loader.cpp, through its host-only inspection hook) against an independent
    Python reader (oracle/vienna2x.py) on synthetic v2.0 files: every table, INF / DEF tokens, comments, skipped enthalpies;
  * how the loader composes what the kernels read under RH_VIENNA_SEM_20: E_ExtLoop ends of a duplex (pf_duplex.c:146,158,
    185,200), exterior / multiloop stems, 1xn and 2x3 loop classes, special hairpins as differences to the plain energy;
  * the install order of RactIP::run (src/ractip.cpp:1563-1567): defaults file -> BL* -> -P file;
  * the Python pf_duplex DP (the checker of the GPU tests) against brute-force enumeration of all duplexes."""
import ctypes
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vienna2x as v2  # noqa: E402

BL = os.path.join(ROOT, "ractip_amd", "data", "vienna_bl_star.params")
KT = v2.KT


@pytest.fixture(scope="module")
def hook(hotlib):
    lib = ctypes.CDLL(os.path.join(ROOT, "ractip_amd", "libractip_hot.so"))
    lib.rh_debug_vienna_value.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int] + [ctypes.c_int] * 5 + [
        ctypes.POINTER(ctypes.c_double)]
    lib.rh_debug_vienna_value.restype = ctypes.c_int

    def get(table, i=0, j=0, k=0, l=0, defaults=None, use_bl=False, param=None, sem=0):
        out = ctypes.c_double()
        rc = lib.rh_debug_vienna_value(defaults.encode() if defaults else None, 1 if use_bl else 0, BL.encode(), param.encode() if param else None,
                                       sem, table, i, j, k, l, ctypes.byref(out))
        assert rc == 0, (rc, table, i, j, k, l)
        return out.value
    return get


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    T = v2.random_tables(11)
    path = str(tmp_path_factory.mktemp("par") / "synthetic_v20.par")
    v2.write_par_v20(path, T)
    return T, path


def w(E):
    return -E * 10.0 / KT


def test_python_reader_round_trips_the_synthetic_file(synth):
    T, path = synth
    R = v2.read_par(path)
    assert R["v20"]
    for k, v in T.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(R[k], v), k
        else:
            assert R[k] == v, k


def test_product_loader_reads_every_table_of_a_v20_file(hook, synth):
    T, path = synth
    rng = np.random.default_rng(5)
    assert hook(22, param=path) == 2                                      # a v2.0 file selects the 2.x semantics
    for tab, key in ((10, "mismatchI"), (11, "mismatchH"), (12, "mismatch1nI"), (13, "mismatch23I")):
        for t, a, b in rng.integers([1, 0, 0], [8, 5, 5], (40, 3)):
            assert hook(tab, t, a, b, param=path) == pytest.approx(w(T[key][t, a, b]), abs=1e-12), (key, t, a, b)
    for t, t2 in rng.integers(1, 8, (30, 2)):
        assert hook(24, t, t2, param=path) == pytest.approx(w(T["stack"][t, t2]), abs=1e-12)
        a, b = rng.integers(0, 5, 2)
        assert hook(25, t, t2, a, b, param=path) == pytest.approx(w(T["int11"][t, t2, a, b]), abs=1e-12)
        assert hook(23, 7, t, t2, param=path) == pytest.approx(w(T["bulge"][1] + T["stack"][t, t2]), abs=1e-12)
    for t, a in rng.integers([1, 0], [8, 5], (20, 2)):                    # dangles are clipped to <= 0 (scale_parameters)
        assert hook(26, t, a, param=path) == pytest.approx(w(min(T["dangle5"][t, a], 0)), abs=1e-12)
        assert hook(27, t, a, param=path) == pytest.approx(w(min(T["dangle3"][t, a], 0)), abs=1e-12)
    assert hook(23, 0, param=path) == pytest.approx(w(T["TerminalAU"]))
    assert hook(23, 1, param=path) == pytest.approx(w(T["DuplexInit"]))
    assert hook(23, 2, param=path) == pytest.approx(w(T["ML_closing"] + T["ML_intern"]))
    assert hook(23, 3, param=path) == pytest.approx(w(T["ML_intern"]))
    assert hook(23, 4, param=path) == pytest.approx(w(T["ML_base"]))
    assert hook(23, 5, param=path) == pytest.approx(T["lxc"] * 10.0 / KT)
    for u in range(3, 31):                                                # the enthalpy sections (poison values) were skipped
        assert hook(23, 6, u, param=path) == pytest.approx(w(T["hairpin"][u]))


def test_loop_classes_and_length_terms_of_both_semantics(hook, synth):
    T, path = synth
    shapes = [(l1, l2) for l1 in range(31) for l2 in range(31 - l1)]
    got = {sem: {sh: (hook(18, sh[0], sh[1], param=path, sem=sem), hook(17, sh[0], sh[1], param=path, sem=sem)) for sh in shapes} for sem in (2, 1)}
    for (l1, l2) in shapes:
        nl, ns = max(l1, l2), min(l1, l2)
        if nl <= 2 and not (ns == 0 and nl == 2):
            assert got[2][l1, l2][0] == 0 and got[1][l1, l2][0] == 0
            continue
        asym = min(T["max_ninio"], (nl - ns) * T["ninio"])
        if ns == 0:
            want20 = want18 = (2, T["bulge"][nl])
        elif ns == 1:                      # 1xn, n >= 3: internal_loop[n+1] + ninio, mismatch1nI (2.x); a generic loop (1.8)
            want20, want18 = (3, T["interior"][nl + 1] + asym), (1, T["interior"][nl + 1] + asym)
        elif ns == 2 and nl == 3:
            want20, want18 = (4, T["interior"][5] + T["ninio"]), (1, T["interior"][5] + asym)
        else:
            want20 = want18 = (1, T["interior"][l1 + l2] + asym)
        assert (got[2][l1, l2][0], got[1][l1, l2][0]) == (want20[0], want18[0]), (l1, l2)
        assert got[2][l1, l2][1] == pytest.approx(w(want20[1])), (l1, l2)
        assert got[1][l1, l2][1] == pytest.approx(w(want18[1])), (l1, l2)


def test_duplex_ends_follow_E_ExtLoop_under_2x_and_dangle_sums_under_18(hook, synth):
    T, path = synth
    P = v2.scaled(T)
    cells = [(t, a, b) for t in range(1, 7) for a in range(6) for b in range(6)]
    got = {sem: {c: hook(14, *c, param=path, sem=sem) for c in cells} for sem in (2, 1)}
    for (t, a, b) in cells:
        si1, sj1 = (a if a < 5 else -1), (b if b < 5 else -1)
        assert got[2][t, a, b] == pytest.approx(w(v2.E_ExtLoop(P, t, si1, sj1)), abs=1e-12), (t, a, b)
        e18 = (P["dangle5"][t, a] if a < 5 else 0) + (P["dangle3"][t, b] if b < 5 else 0) + (T["TerminalAU"] if t > 2 else 0)
        assert got[1][t, a, b] == pytest.approx(w(e18), abs=1e-12), (t, a, b)


def test_stems_of_exterior_and_multi_loops(hook, synth):
    T, path = synth
    cells = [(t, a, b) for t in range(1, 7) for a in range(5) for b in range(5)]
    got = {sem: {c: (hook(15, *c, param=path, sem=sem), hook(16, *c, param=path, sem=sem)) for c in cells} for sem in (2, 1)}
    for (t, a, b) in cells:
        assert got[2][t, a, b][0] == pytest.approx(v2.w_stem(T, "mismatchExt", t, a, b), abs=1e-12)
        assert got[2][t, a, b][1] == pytest.approx(v2.w_stem(T, "mismatchM", t, a, b), abs=1e-12)
        d = (v2.smooth_w(T["dangle5"][t, a]) if a else 0.0) + (v2.smooth_w(T["dangle3"][t, b]) if b else 0.0) + (w(T["TerminalAU"]) if t > 2 else 0.0)
        assert got[1][t, a, b][0] == pytest.approx(d, abs=1e-12)     # 1.8: smoothed dangles on both sides,
        assert got[1][t, a, b][1] == pytest.approx(d, abs=1e-12)     # the same for both loop kinds


def code(sq):
    c = 0
    for ch in sq:
        c = c * 4 + "ACGU".index(ch)
    return c


def test_special_hairpins_replace_the_energy_under_2x_and_add_to_it_under_18(hook, synth):
    T, path = synth
    for sq, E in T["Tetraloops"].items():
        t = v2.PAIR["ACGU".index(sq[0]) + 1, "ACGU".index(sq[-1]) + 1]
        plain = T["hairpin"][4] + T["mismatchH"][t, "ACGU".index(sq[1]) + 1, "ACGU".index(sq[-2]) + 1]
        assert hook(19, code(sq), param=path) == pytest.approx(w(E) - w(plain))
    for sq, E in T["Triloops"].items():
        t = v2.PAIR["ACGU".index(sq[0]) + 1, "ACGU".index(sq[-1]) + 1]
        plain = T["hairpin"][3] + (T["TerminalAU"] if t > 2 else 0)
        assert hook(20, code(sq), param=path) == pytest.approx(w(E) - w(plain))
    for sq, E in T["Hexaloops"].items():
        t = v2.PAIR["ACGU".index(sq[0]) + 1, "ACGU".index(sq[-1]) + 1]
        plain = T["hairpin"][6] + T["mismatchH"][t, "ACGU".index(sq[1]) + 1, "ACGU".index(sq[-2]) + 1]
        assert hook(21, code(sq), param=path) == pytest.approx(w(E) - w(plain))
    assert hook(21, code("AAAAAAAA"), param=path) == 0.0
    # the bundled BL* file: tetraloop BONUS energies (copy_Tetra_loop, boltzmann_param.c:6025), 1.8 semantics
    assert hook(22, param=BL) == 1
    bonus = [hook(19, c, param=BL) for c in range(4096)]
    assert sum(1 for b in bonus if b != 0.0) >= 20


def test_install_order_defaults_then_bl_then_param_file(hook, synth, tmp_path):
    T, path = synth
    # defaults = the synthetic v2.0 file, BL* on top: the tables BL* holds win, the 2.x-only tables survive from the defaults
    assert hook(22, defaults=path, use_bl=True) == 2
    assert hook(12, 3, 1, 2, defaults=path, use_bl=True) == pytest.approx(w(T["mismatch1nI"][3, 1, 2]))
    assert hook(24, 1, 2, defaults=path, use_bl=True) == pytest.approx(hook(24, 1, 2, param=BL))
    assert hook(24, 1, 2, defaults=path, use_bl=True) != pytest.approx(w(T["stack"][1, 2]))
    # a partial -P file on top of BL*: only its sections change; DEF keeps the value
    part = tmp_path / "partial.par"
    part.write_text("## RNAfold parameter file v2.0\n\n# mismatch_exterior\n" + "\n".join(" ".join(["-70"] * 5) for _ in range(35)) +
                    "\n\n# NINIO\n DEF 0 250\n\n#END\n")
    assert hook(24, 1, 2, use_bl=True, param=str(part)) == pytest.approx(hook(24, 1, 2, param=BL))
    assert hook(14, 1, 2, 3, use_bl=True, param=str(part)) == pytest.approx(w(-70))
    # BL* alone under the 2.x semantics: the 2.x-only tables are zero (nothing provides them)
    assert hook(12, 3, 1, 2, use_bl=True, sem=2) == 0.0 and hook(14, 1, 2, 3, use_bl=True, sem=2) == 0.0
    assert hook(14, 5, 2, 3, use_bl=True, sem=2) == pytest.approx(hook(23, 0, param=BL))      # TerminalAU only


def test_a_1x_layout_file_keeps_the_18_semantics(hook, tmp_path):
    f = tmp_path / "old.par"
    f.write_text("## RNAfold parameter file\n\n# stack_energies\n/*          CG     GC     GU     UG     AU     UA  */\n" +
                 "\n".join(" ".join(str(-100 - 10 * i - j) for j in range(7)) for i in range(7)) +
                 "\n\n# ML_params\n/* cu cc ci TerminalAU */\n 0 340 40 50\n\n# NINIO\n/* m max */\n 50 300\n\n# Tetraloops\nGGGGAC -300\n" +
                 # the loader refuses a model in which a table the kernels read comes from no source: the rest of the core tables, all zero
                 "".join("\n# %s\n%s\n" % (name, " ".join(["0"] * cnt)) for name, cnt in (
                     ("mismatch_interior", 175), ("mismatch_hairpin", 175), ("dangle5", 35), ("dangle3", 35), ("int11_energies", 1225),
                     ("int21_energies", 6125), ("int22_energies", 9216), ("bulge", 31), ("internal_loop", 31), ("hairpin", 31))) +
                 "\n# END\n")
    assert hook(22, param=str(f)) == 1
    assert hook(24, 2, 3, param=str(f)) == pytest.approx(w(-100 - 10 * 1 - 2))
    assert hook(23, 0, param=str(f)) == pytest.approx(w(50))
    assert hook(19, code("GGGGAC"), param=str(f)) == pytest.approx(w(-300))
    assert hook(17, 1, 4, param=str(f)) == pytest.approx(w(0 + min(300, 3 * 50)))


def test_loader_reports_malformed_files(hotlib, tmp_path):
    lib = ctypes.CDLL(os.path.join(ROOT, "ractip_amd", "libractip_hot.so"))
    lib.rh_debug_vienna_value.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int] + [ctypes.c_int] * 5 + [
        ctypes.POINTER(ctypes.c_double)]
    out = ctypes.c_double()
    bad = tmp_path / "bad.par"
    bad.write_text("## RNAfold parameter file v2.0\n\n# stack\n 1 2 3\n\n#END\n")
    assert lib.rh_debug_vienna_value(None, 0, BL.encode(), str(bad).encode(), 0, 22, 0, 0, 0, 0, ctypes.byref(out)) == -4
    assert lib.rh_debug_vienna_value(None, 0, BL.encode(), b"/nonexistent.par", 0, 22, 0, 0, 0, 0, ctypes.byref(out)) == -4
    assert lib.rh_debug_vienna_value(None, 1, BL.encode(), None, 7, 22, 0, 0, 0, 0, ctypes.byref(out)) == -4   # unknown semantics
    # a file that leaves core tables to no source at all (no defaults, no BL*) is refused instead of computing with zero energies ...
    part = tmp_path / "partial.par"
    part.write_text("## RNAfold parameter file\n\n# stack_energies\n" + "\n".join(" ".join(["-100"] * 7) for _ in range(7)) + "\n\n# END\n")
    assert lib.rh_debug_vienna_value(None, 0, BL.encode(), str(part).encode(), 0, 22, 0, 0, 0, 0, ctypes.byref(out)) == -4
    # ... and accepted on top of the BL* tables, which supply the rest (install order of src/ractip.cpp:1563-1567)
    assert lib.rh_debug_vienna_value(None, 1, BL.encode(), str(part).encode(), 0, 22, 0, 0, 0, 0, ctypes.byref(out)) == 0
    # a truncated flat dump (a table cut out of the bundled BL* file) as the only source is refused too
    lines = open(BL).read().splitlines()
    k0 = next(k for k, l in enumerate(lines) if l.startswith("bulge37"))
    k1 = next(k for k in range(k0 + 1, len(lines)) if lines[k] and not lines[k][0].isdigit() and lines[k][0] not in "-# ")
    cut = tmp_path / "cut.params"
    cut.write_text("\n".join(lines[:k0] + lines[k1:]) + "\n")
    assert lib.rh_debug_vienna_value(None, 1, str(cut).encode(), None, 0, 22, 0, 0, 0, 0, ctypes.byref(out)) == -4


@pytest.mark.parametrize("seed", range(4))
def test_python_pf_duplex_matches_enumeration_of_all_duplexes(seed):
    rng = np.random.default_rng(100 + seed)
    T = v2.random_tables(200 + seed)
    s1 = "".join("ACGU"[k] for k in rng.integers(0, 4, 6))
    s2 = "".join("ACGU"[k] for k in rng.integers(0, 4, 6))
    efw, ebk, pr = v2.pf_duplex(T, s1, s2)
    lz, prb = v2.brute_duplex(T, s1, s2)
    if lz == -math.inf:
        assert efw == -math.inf
        return
    assert efw == pytest.approx(lz, abs=1e-9) and ebk == pytest.approx(lz, abs=1e-9)
    assert np.allclose(pr, prb, atol=1e-10)
