"""Joint structures: the integer programme of RactIP::solve (ractip_amd/ilp.py) over the probability matrices.

CPU part: the solver wrapper, and the only program output the reference records -- README.md:91-97, `ractip DIS.fa DIS.fa`
on its default path (ViennaRNA pf_fold + pf_unstru + co_pf_fold, unknown version) -- as a smoke check of the
parity-unpinned Vienna-BL restatement.  GPU part (BASELINE.json: "identical bracket structures on the bundled data/*.fa
pairs"): for all eight interacting pairs of /root/reference/data the structure decoded from the HIP path's matrices equals
the one decoded from the reference engines' matrices (golden vectors; CONTRAfold model) resp. from the CPU restatement
(Vienna-BL model), through the same programme."""
import numpy as np
import pytest

from ractip_amd import ilp

PAIRS = [("DIS", "DIS"), ("CopA", "CopT"), ("IncRNA54", "RepZ"), ("MicA", "ompA"), ("OxyS", "fhlA"), ("R1inv", "R2inv"),
         ("RyhB", "SodB"), ("Tar", "Tarstar")]
README_DIS = "((((.(((((((..[[[[[[.)))))))...))))"   # README.md:94, second strand with ']' for '['


def balanced(r):
    depth = 0
    for ch in r:
        depth += ch == "("
        depth -= ch == ")"
        assert depth >= 0
    return depth == 0


def test_ip_model_is_class_ip():
    ip = ilp.IPModel()
    a, b, c = ip.make_variable(1.0), ip.make_variable(2.0), ip.make_variable(-1.0)
    r = ip.make_constraint(ilp.UP, 0, 1)           # a + b <= 1
    ip.add_constraint(r, a, 1); ip.add_constraint(r, b, 1)
    r = ip.make_constraint(ilp.LO, 1, 0)           # b + c >= 1
    ip.add_constraint(r, b, 1); ip.add_constraint(r, c, 1)
    assert abs(ip.solve() - 2.0) < 1e-9
    assert [round(ip.get_value(k)) for k in (a, b, c)] == [0, 1, 0]


def test_readme_dis_dis_structure_needs_a_looser_accessibility_threshold(golden):
    """The Vienna-BL restatement + programme reproduces the README structure (README.md:93-97) only with --acc-th 0.01."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    s = str(golden["mc/DIS/seq"])
    f, co = vo.mccaskill(s, max_w=15), vo.cofold(s, s)
    r1, r2, _ = ilp.solve(s, s, f["post"], f["post"], co["hp"], f["up"], f["up"], ilp.Options(th_ac=0.01))
    assert r1 == README_DIS and r2 == README_DIS.replace("[", "]")
    assert 0.003 < f["up"][10][12] < 0.01   # P(letters 11..23 unpaired) = 0.0038: 0.14 kcal/mol above the default threshold


@pytest.mark.xfail(strict=True, reason="KNOWN GAP, parity unpinned: at RactIP's default thresholds the only output the reference "
                   "records for its default path (README.md:91-97, `ractip DIS.fa DIS.fa`, unknown ViennaRNA 2.x build) is NOT "
                   "reproduced -- the interaction comes out two pairs longer on each side.  The Vienna-BL model here has "
                   "ViennaRNA-1.8 loop-energy semantics (the 1.8 branch of src/pf_duplex.c:209-433); CMakeLists.txt:28 builds the "
                   "2.x branch.  This test turns green only when the default-flag output matches the README.")
def test_readme_dis_dis_default_thresholds_match_the_reference_output(golden):
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    s = str(golden["mc/DIS/seq"])
    f, co = vo.mccaskill(s, max_w=15), vo.cofold(s, s)
    r1, r2, _ = ilp.solve(s, s, f["post"], f["post"], co["hp"], f["up"], f["up"])   # cmdline.c defaults: -a 0.7 -t 0.5, th_ac 0.003
    assert r1 == README_DIS and r2 == README_DIS.replace("[", "]")


def test_structures_from_reference_matrices_are_well_formed(golden):
    for a, b in PAIRS:
        s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
        hp = golden["dx/%s+%s/post" % (a, b)].reshape(len(s1) + 1, len(s2) + 1)
        r1, r2, _ = ilp.solve(s1, s2, golden["mc/%s/post" % a], golden["mc/%s/post" % b], hp, None, None, ilp.Options(min_w=0))
        assert len(r1) == len(s1) and len(r2) == len(s2) and balanced(r1) and balanced(r2)
        assert r1.count("[") == r2.count("]")


def test_rip_tables_round_trip_and_give_the_same_structure(golden, tmp_path):
    """RIP text tables (src/ractip.cpp:461-514): write -> read (float, sequence 2 reversed in the file) -> same matrices,
    hence the same programme and structure a stock `ractip --rip FILE --min-w 0` would build from them."""
    from ractip_amd import rip
    a, b = "CopA", "CopT"
    s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
    bp1, bp2 = golden["mc/%s/post" % a], golden["mc/%s/post" % b]
    hp = golden["dx/%s+%s/post" % (a, b)].reshape(len(s1) + 1, len(s2) + 1)
    path = str(tmp_path / "pair.rip")
    rip.write_rip(path, s1, s2, bp1, bp2, hp)
    r1, r2, r3 = rip.read_rip(path, s1, s2)
    assert np.array_equal(r1, bp1.astype(np.float32)) and np.array_equal(r2, bp2.astype(np.float32))
    assert np.array_equal(r3, hp.astype(np.float32))
    opt = ilp.Options(min_w=0)
    assert ilp.solve(s1, s2, r1, r2, r3, None, None, opt)[:2] == ilp.solve(s1, s2, bp1, bp2, hp, None, None, opt)[:2]
    # the reversal convention of table S: the file's (i, j) is s2's pair (L2-j+1, L2-i+1)
    lines = open(path).read().split("Table S:")[1].split("Table I:")[0].split()
    i, j, p = int(lines[0]), int(lines[1]), float(lines[2])
    n2 = len(s2)
    x, y = n2 - j + 1, n2 - i + 1
    assert x < y and abs(bp2[x * (2 * (n2 + 1) - x - 1) // 2 + y] - p) < 1e-15


def _oracle_matrices(pairs):
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    out = []
    for a, b in pairs:
        fa, fb, co = vo.mccaskill(a, max_w=15), vo.mccaskill(b, max_w=15), vo.cofold(a, b)
        out.append(dict(bp1=fa["post"], bp2=fb["post"], up1=fa["up"], up2=fb["up"], hp=co["hp"]))
    return out


def test_zscore_loop_on_cpu_matrices(golden):
    """The z-score loop (src/ractip.cpp:1624-1670) end to end over the CPU restatement's matrices: energies are finite,
    the shuffles keep the dinucleotide content, the statistic is reproducible."""
    from ractip_amd import pipeline
    s1, s2 = str(golden["mc/Tar/seq"]), str(golden["mc/Tarstar/seq"])
    a = pipeline.zscore(s1, s2, mode=12, num_shuffling=4, seed=1, matrices=_oracle_matrices)
    b = pipeline.zscore(s1, s2, mode=12, num_shuffling=4, seed=1, matrices=_oracle_matrices)
    assert a == b and a["r1"].count("[") == a["r2"].count("]") > 0
    assert all(np.isfinite(a[k]) for k in ("e1", "e2", "e3", "e1s", "e2s")) and a["e3"] < 0
    assert np.isfinite(a["zscore"]) and a["zscore"] < 0      # the native Tar/Tar* kissing complex beats its shuffles


def test_zscore_loop_is_finite_on_oxys_fhla(golden):
    """BASELINE config 5's pair: shuffles whose single-sequence programme yields crossing pairs (bracket strings that match
    unpairable letters) must still give finite energies, as they do under ViennaRNA's evaluator."""
    from ractip_amd import pipeline
    import os
    fa = [l.strip() for l in open(os.path.join(os.path.dirname(pipeline.__file__), "data", "config5_OxyS_fhlA.fa")) if not l.startswith(">")]
    z = pipeline.zscore(fa[0], fa[1], mode=12, num_shuffling=3, seed=1, matrices=_oracle_matrices)
    assert all(np.isfinite(z[k]) for k in ("e1", "e2", "e3", "e1s", "e2s", "zscore", "zscore_s")), z


@pytest.mark.gpu
def test_zscore_loop_gpu_equals_cpu_matrices(hotlib, golden):
    from ractip_amd import pipeline
    s1, s2 = str(golden["mc/DIS/seq"]), str(golden["mc/Tar/seq"])
    g = pipeline.zscore(s1, s2, mode=12, num_shuffling=5, seed=1)
    c = pipeline.zscore(s1, s2, mode=12, num_shuffling=5, seed=1, matrices=_oracle_matrices)
    assert g == c


@pytest.mark.gpu
def test_identical_joint_structures_on_all_bundled_pairs(hotlib, golden):
    import ractip_amd
    from ractip_amd import pipeline
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    cf = ractip_amd.Context(device=0)
    vi = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
    try:
        for a, b in PAIRS:
            s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
            # CONTRAfold model: reference engines (golden vectors) vs HIP path
            hp = golden["dx/%s+%s/post" % (a, b)].reshape(len(s1) + 1, len(s2) + 1)
            want = ilp.solve(s1, s2, golden["mc/%s/post" % a], golden["mc/%s/post" % b], hp, None, None, ilp.Options(min_w=0))[:2]
            got = pipeline.predict(s1, s2, model="contrafold", ctx=cf)[:2]
            assert got == want, (a, b, got, want)
            # Vienna-BL model, default path (rnafold + co_pf_fold): CPU restatement vs HIP path
            f1, f2, co = vo.mccaskill(s1, max_w=15), vo.mccaskill(s2, max_w=15), vo.cofold(s1, s2)
            want = ilp.solve(s1, s2, f1["post"], f2["post"], co["hp"], f1["up"], f2["up"])[:2]
            got = pipeline.predict(s1, s2, model="vienna", ctx=vi)[:2]
            assert got == want, (a, b, got, want)
            # --duplex
            want = ilp.solve(s1, s2, f1["post"], f2["post"], vo.pf_duplex(s1, s2)["pr"], f1["up"], f2["up"])[:2]
            got = pipeline.predict(s1, s2, model="vienna", duplex=True, ctx=vi)[:2]
            assert got == want, (a, b, got, want)
    finally:
        cf.close()
        vi.close()
