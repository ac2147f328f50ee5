"""B_alg: the numpy counter in ractip_amd/balg.py must equal the
load/store counters of the instrumented CPU oracle, and sit near the closed form
BASELINE.md quotes for uniform-random sequences."""
import numpy as np

from ractip_amd import balg
from ractip_amd.seqgen import random_pair


def test_counts_match_instrumented_oracle(oracle):
    rng = np.random.RandomState(5)
    for n in (1, 2, 3, 5, 9, 33, 47, 80):
        seq = "".join(rng.choice(list("ACGU"), n))
        loads, stores = oracle.count(oracle.inference, seq)
        c = balg.mccaskill_counts(seq)
        assert loads == sum(v[0] for v in c.values()), n
        assert stores == sum(v[1] for v in c.values()), n
    for n1, n2 in ((1, 1), (4, 9), (40, 33), (35, 80)):
        s1 = "".join(rng.choice(list("ACGU"), n1))
        s2 = "".join(rng.choice(list("ACGU"), n2))
        loads, stores = oracle.count(oracle.duplex, s1, s2)
        c = balg.duplex_counts(s1, s2)
        assert loads == sum(v[0] for v in c.values()), (n1, n2)
        assert stores == sum(v[1] for v in c.values()), (n1, n2)


def test_n500_bytes_near_baseline_closed_form():
    s1, s2 = random_pair(500)
    b = balg.pair_bytes(s1, s2)
    # ~1.77 GB per sequence, ~0.43 GB duplex, ~3.97 GB per pair
    assert 3.3e9 < b["total"] < 4.6e9, b
