"""CPU suite for the PARITY-UNPINNED Vienna-BL duplex restatement (oracle/vienna_oracle.c): no reference output
exists for this path (ViennaRNA is absent and unversioned), so the checks are model-independent -- the DP equals
brute-force enumeration of every duplex under the same energy function, and forward log Z == backward log Z."""
import numpy as np
import pytest

from _oracle import ViennaOracle


@pytest.fixture(scope="module")
def vo():
    return ViennaOracle()


def test_dp_equals_bruteforce_enumeration(vo):
    rng = np.random.RandomState(12)
    for n1, n2 in ((1, 1), (2, 3), (4, 4), (5, 7), (8, 6), (9, 9)):
        for _ in range(3):
            s1 = "".join(rng.choice(list("ACGU"), n1))
            s2 = "".join(rng.choice(list("ACGU"), n2))
            a, b = vo.pf_duplex(s1, s2), vo.bruteforce(s1, s2)
            if not np.isfinite(b["logZ"]):
                assert not np.isfinite(a["logZ"]) and a["pr"].max() == 0
                continue
            assert abs(a["logZ"] - b["logZ"]) < 1e-10, (s1, s2)
            assert np.abs(a["pr"] - b["pr"]).max() < 1e-10, (s1, s2)


def test_forward_equals_backward_and_probabilities_are_sane(vo, golden):
    for a, b in (("DIS", "DIS"), ("CopA", "CopT"), ("OxyS", "fhlA"), ("Tar", "Tarstar")):
        s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
        r = vo.pf_duplex(s1, s2)
        assert abs(r["logZ"] - r["logZ_bk"]) < 1e-9 * max(1.0, abs(r["logZ"]))
        pr = r["pr"]
        assert pr.min() >= 0 and pr.max() <= 1 + 1e-12
        assert pr.sum(axis=1).max() <= 1 + 1e-9 and pr.sum(axis=0).max() <= 1 + 1e-9
    # CopA/CopT are fully complementary antisense RNAs: the model must find the full-length duplex
    r = vo.pf_duplex(str(golden["mc/CopA/seq"]), str(golden["mc/CopT/seq"]))
    assert r["pr"].max() > 0.99


def test_t_reads_as_u_and_unknown_letters_do_not_pair(vo):
    a = vo.pf_duplex("GGGAAACCC", "GGGTTTCCC")
    b = vo.pf_duplex("GGGAAACCC", "GGGUUUCCC")
    assert a["logZ"] == b["logZ"]
    assert vo.pf_duplex("NNNN", "NNNN")["pr"].max() == 0


# ---- McCaskill / accessibility restatement (pf_fold + pf_unstru semantics, PARITY UNPINNED) ---------------------
MULTILOOP_SEQS = [
    "GGGAAACCCAGGGAAACCCA",            # two hairpins, multiloop (1,19)
    "GGAAACCCAGGGAAACCC",              # three-branch multiloop with a trailing unpaired letter
    "GGCGAAAGCCAAGGCGAAAGCCAA",        # tetraloop bonus CGAAAG
    "GGGGACGAAAGUCCUUGGGAGACUCCC",
    "GCGCGAAAGCAUGCGAAAGCAGCAAGC",
]


def test_mccaskill_equals_bruteforce_enumeration(vo):
    """bp, log Z and P(region unpaired) of the DP == explicit enumeration of every secondary structure (exact
    marginals of the Boltzmann ensemble are what pf_fold/pf_unstru define)."""
    rng = np.random.RandomState(3)
    seqs = list(MULTILOOP_SEQS)
    seqs += ["".join(rng.choice(list("ACGU"), n, p=[.15, .35, .35, .15])) for n in (1, 4, 5, 9, 14, 18, 20, 22, 24, 26)]
    for s in seqs:
        a, b = vo.mccaskill(s, max_w=10), vo.fold_bruteforce(s, max_w=10)
        assert abs(a["logZ"] - b["logZ"]) < 1e-11, s
        assert abs(a["logZ"] - a["logZ_out"]) < 1e-11, s
        assert np.abs(a["post"] - b["post"]).max() < 1e-11, s
        assert np.abs(a["up"] - b["up"]).max() < 1e-11, s


def test_mccaskill_invariants_on_bundled_sequences(vo, golden):
    for name in ("DIS", "CopA", "CopT", "OxyS", "fhlA"):
        s = str(golden["mc/%s/seq" % name])
        n = len(s)
        r = vo.mccaskill(s, max_w=15)
        assert abs(r["logZ"] - r["logZ_out"]) < 1e-9 * max(1.0, abs(r["logZ"]))
        bp, up = r["post"], r["up"]
        assert bp.min() >= 0 and bp.max() <= 1 + 1e-12
        # width-1 accessibility + pairing probabilities of a letter sum to one
        paired = np.zeros(n + 1)
        for i in range(1, n + 1):
            off = i * (2 * (n + 1) - i - 1) // 2
            for j in range(i + 1, n + 1):
                paired[i] += bp[off + j]
                paired[j] += bp[off + j]
        assert np.abs(paired[1:] + up[:, 0] - 1).max() < 1e-9
        # wider regions are less accessible; regions running off the end are 0
        assert (np.diff(up, axis=1) <= 1e-12).all()
        assert up[n - 1, 1] == 0 and up[n - 3, 3] == 0


# ---- two-molecule ensemble (co_pf_fold semantics, src/ractip.cpp:400-458; PARITY UNPINNED) ------------------------
def test_cofold_equals_bruteforce_enumeration(vo):
    """log Z, the full pair matrix of s1+s2 and hence hp == explicit enumeration of every joint structure, where the
    loop that holds the gap between the molecules is exterior-like and no dangle crosses it."""
    rng = np.random.RandomState(11)
    cases = [("GGGAAACCC", "GGGUUUCCC"), ("GGGG", "CCCC"), ("G", "C"), ("GGGAC", "GUCCC"), ("GCGCAAAGCGC", "GGGAAACCCA"),
             ("GGGAAACCCAGG", "CCUGGGAAACCC"), ("GGAGGAAACUCC", "GGAGGAAACUCC"), ("GGGAAACCCAGGGAAACCCA", "UGGG"),
             ("CCCA", "GGGAAACCCAGGGAAACCCAUGGG")]
    for n1, n2 in ((3, 5), (6, 6), (8, 9), (12, 10), (11, 13), (13, 12), (5, 20), (14, 13)):
        for _ in range(2):
            cases.append(("".join(rng.choice(list("ACGU"), n1, p=[.15, .35, .35, .15])),
                          "".join(rng.choice(list("ACGU"), n2, p=[.15, .35, .35, .15]))))
    for s1, s2 in cases:
        a, b = vo.cofold(s1, s2), vo.cofold(s1, s2, bruteforce=True)
        assert abs(a["logZ"] - b["logZ"]) < 1e-11, (s1, s2)
        assert abs(a["logZ"] - a["logZ_out"]) < 1e-11, (s1, s2)
        assert np.abs(a["post"] - b["post"]).max() < 1e-11, (s1, s2)


def test_cofold_limits(vo, golden):
    # a molecule that cannot pair with its partner folds as if alone: Z factorises
    a, b = "GGGAAACCCAGGGAAACCCA", "AAAAAAAA"
    zab = vo.cofold(a, b)["logZ"]
    assert abs(zab - vo.mccaskill(a)["logZ"] - vo.mccaskill(b)["logZ"]) < 1e-10
    # bundled pair: inside == outside, probabilities sane, hp rows/columns are distributions
    s1, s2 = str(golden["mc/CopA/seq"]), str(golden["mc/CopT/seq"])
    r = vo.cofold(s1, s2)
    assert abs(r["logZ"] - r["logZ_out"]) < 1e-9 * abs(r["logZ"])
    hp = r["hp"]
    assert hp.min() >= 0 and hp.sum(axis=1).max() <= 1 + 1e-9 and hp.sum(axis=0).max() <= 1 + 1e-9
    assert hp.max() > 0.9     # fully complementary antisense pair


def test_structure_constraints_equal_bruteforce(vo):
    """fold_constrained (src/ractip.cpp:271-291): the DP under the allowed-pair mask of ViennaRNA-1.8 make_ptypes == the
    enumeration of every structure that respects the mask."""
    s = "GGGAAACCCAGGGAAACCCA"
    for c in ("xxx.................", "(......)............", "<<<......>>>........", ".(......)......x....",
              "((....))..(((...))).", ".|||....x...........", "(((...)))"):
        a, b = vo.mccaskill(s, max_w=6, constraint=c), vo.fold_bruteforce(s, max_w=6, constraint=c)
        assert abs(a["logZ"] - b["logZ"]) < 1e-11 and np.abs(a["post"] - b["post"]).max() < 1e-11, c
        assert np.abs(a["up"] - b["up"]).max() < 1e-11, c
    # a forced pair has the probability of "letter 1 is paired at all" and nothing crosses it
    a = vo.mccaskill(s, constraint="(......)")
    n = len(s)
    off1 = 1 * (2 * (n + 1) - 1 - 1) // 2
    assert a["post"][off1 + 2:off1 + n + 1].sum() == a["post"][off1 + 8]
