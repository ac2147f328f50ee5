"""Structure energies of the Vienna-BL model (ractip_amd/energy.py, the caller-side energy_of_structure / energy_of_duplex of
src/ractip.cpp:1254, 1299, 1457, 1529-1559; PARITY UNPINNED).  Check: summing exp(-E/kT) over EVERY structure gives the
partition function of the probability layer (its CPU restatement), with and without a cut point."""
import itertools
import math

import numpy as np
import pytest

from ractip_amd import energy


def all_structures(n, min_dist=4):
    """Every non-crossing pair set on n letters with pairs at least min_dist apart, as bracket strings."""
    def rec(i, j):
        if j - i + 1 <= 0:
            yield ""
            return
        for rest in rec(i + 1, j):
            yield "." + rest
        for k in range(i + min_dist, j + 1):
            for inner in rec(i + 1, k - 1):
                for rest in rec(k + 1, j):
                    yield "(" + inner + ")" + rest
    return rec(1, n)


@pytest.fixture(scope="module")
def vo():
    from _oracle import ViennaOracle
    return ViennaOracle()


def test_sum_over_structures_is_the_partition_function(vo):
    rng = np.random.RandomState(2)
    seqs = ["GGGAAACCC", "GGCGAAAGCC", "GGGAAACCCAGG"] + ["".join(rng.choice(list("ACGU"), n, p=[.15, .35, .35, .15])) for n in (8, 11, 13)]
    kT = energy.KT / 1000.0
    for s in seqs:
        z = sum(math.exp(-e / kT) for e in (energy.energy_of_structure(s, st) for st in all_structures(len(s))) if math.isfinite(e))
        assert abs(math.log(z) - vo.mccaskill(s)["logZ"]) < 1e-10, s
    for s1, s2 in (("GGGAC", "GUCCC"), ("GGGAAACC", "GGUCCC"), ("GCGCA", "UGCGCAA")):
        n1 = len(s1)
        z = sum(math.exp(-e / kT) for e in (energy.energy_of_structure(s1 + s2, st, cut=n1) for st in all_structures(len(s1 + s2)))
                if math.isfinite(e))
        assert abs(math.log(z) - vo.cofold(s1, s2)["logZ"]) < 1e-10, (s1, s2)


def test_energy_of_duplex_and_known_values():
    # the open chain has energy 0; a GC helix with a GAAA tetraloop is strongly negative
    assert energy.energy_of_structure("GGGAAACCC", ".........") == 0.0
    assert energy.energy_of_structure("GGCGAAAGCC", "(((....)))") < -2.0
    # energy_of_duplex: internal pairs are erased, brackets become pairs across the cut
    s1, s2 = "GGGAAACCC", "GGGUUUCCC"
    e = energy.energy_of_duplex(s1, s2, "[[[...(.)", "(.)...]]]")
    assert e == energy.energy_of_structure(s1 + s2, "(((............)))", cut=len(s1))
    assert math.isinf(energy.energy_of_structure("AAAAAAAA", "(......)"))


def test_noncanonical_pairs_score_as_type_7():
    """ViennaRNA's evaluator scores letters that cannot pair as pair type 7 instead of rejecting the structure; the z-score
    loop relies on it (its single-sequence programme has no crossing constraint, src/ractip.cpp:1355-1465)."""
    assert math.isinf(energy.energy_of_structure("AAAAAAAA", "(......)"))
    e = energy.energy_of_structure("AAAAAAAA", "(......)", noncanonical=True)
    assert math.isfinite(e) and e > 0
    # a canonical structure is unaffected by the switch
    assert energy.energy_of_structure("GGCGAAAGCC", "(((....)))", noncanonical=True) == energy.energy_of_structure("GGCGAAAGCC", "(((....)))")
