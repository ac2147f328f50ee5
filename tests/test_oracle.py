"""CPU suite: the oracle (oracle/cf_oracle.c) against the committed golden vectors
generated from the reference's own engines, and -- where oracle/_ref has been built
in this tree -- directly against those engines."""
import os

import numpy as np
import pytest

from _oracle import NEG, Reference, assert_log_close, assert_prob_close, tri_offset, tri_size


def test_known_answers_from_survey(golden):
    # SURVEY.md section 8c "Known answers" (InferenceEngine<double>, DuplexEngine<double>)
    assert abs(float(golden["mc/DIS/logZ"]) - 6.837353829177) < 1e-11
    assert abs(float(golden["mc/CopA/logZ"]) - 16.125678437320) < 1e-11
    assert abs(float(golden["mc/fhlA/logZ"]) - 20.144171522420) < 1e-11
    assert abs(float(golden["mc500/mt500a/logZ"]) - 60.344965) < 1e-6
    assert abs(float(golden["mc/mt200/logZ"]) - 21.463076) < 1e-6
    assert abs(golden["dx/CopA+CopT/logZ2"][0] - 114.691580657731) < 1e-10
    assert abs(golden["dx/OxyS+fhlA/logZ2"][0] - 136.742436218434) < 1e-10
    post = golden["mc/DIS/post"]
    assert abs(post.sum() - 10.4353823636) < 1e-9 and int((post > 0.5).sum()) == 12
    assert abs(post[tri_offset(35, 10) + 24] - 0.972665538867) < 1e-11


def test_oracle_mccaskill_vs_golden(oracle, golden):
    for nm in golden["mc_names"]:
        seq = str(golden["mc/%s/seq" % nm])
        r = oracle.inference(seq, tables=("mc/%s/tables" % nm) in golden.files)
        assert abs(r["logZ"] - float(golden["mc/%s/logZ" % nm])) < 1e-10, nm
        assert_prob_close(r["post"], golden["mc/%s/post" % nm], rel=1e-9, what="post " + nm)
        assert_log_close(r["f5"], golden["mc/%s/f5" % nm], tol=1e-10, what="f5 " + nm)
        if "tables" in r:
            assert_log_close(r["tables"], golden["mc/%s/tables" % nm], tol=1e-10, what="tables " + nm)


def test_oracle_duplex_vs_golden(oracle, golden):
    for key in golden["dx_names"]:
        a, b = key.split("+")
        s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
        r = oracle.duplex(s1, s2)
        assert_log_close(r["logZ2"], golden["dx/%s/logZ2" % key], tol=1e-10, what="logZ " + key)
        assert_prob_close(r["post"], golden["dx/%s/post" % key], rel=1e-9, what="post " + key)
        if ("dx/%s/inside" % key) in golden.files:
            assert_log_close(r["inside"], golden["dx/%s/inside" % key], tol=1e-10, what="inside " + key)
            assert_log_close(r["outside"], golden["dx/%s/outside" % key], tol=1e-10, what="outside " + key)


def test_oracle_n500_vs_golden(oracle, golden):
    seq = str(golden["mc500/mt500a/seq"])
    r = oracle.inference(seq)
    assert abs(r["logZ"] - float(golden["mc500/mt500a/logZ"])) < 1e-9
    assert abs(r["post"].sum() - float(golden["mc500/mt500a/post_sum"])) < 1e-7
    assert_prob_close(r["post"][golden["mc500/mt500a/idx"]], golden["mc500/mt500a/val"], rel=1e-8, what="post n=500")


def test_oracle_vs_reference_engines_if_built(oracle):
    try:
        ref = Reference()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not built in this tree (reference sources absent)")
    rng = np.random.RandomState(7)
    for n in (1, 2, 6, 9, 23, 51, 77):
        seq = "".join(rng.choice(list("ACGU"), n))
        a, b = oracle.inference(seq), ref.inference(seq)
        assert abs(a["logZ"] - b["logZ"]) < 1e-10
        assert_prob_close(a["post"], b["post"], rel=1e-9, what="n=%d" % n)
    s1 = "".join(rng.choice(list("ACGU"), 37))
    s2 = "".join(rng.choice(list("ACGU"), 58))
    a, b = oracle.duplex(s1, s2), ref.duplex(s1, s2)
    assert_log_close(a["logZ2"], b["logZ2"], tol=1e-10)
    assert_prob_close(a["post"], b["post"], rel=1e-9)


def test_oracle_edge_cases(oracle):
    # nothing can pair: logZ is n * external_unpaired, posterior identically zero
    r = oracle.inference("AAAAAAAA")
    assert r["post"].max() == 0.0 and abs(r["logZ"] - 8 * (-0.0097288309)) < 1e-12
    # unknown letters and T are unpairable (InferenceEngine.ipp:379-384)
    r = oracle.inference("GGGGTTTTNNNN")
    assert r["post"].max() == 0.0
    # inside logZ == outside logZ (F5o[0]) for the McCaskill engine
    r = oracle.inference("GGGAAAUCCCGCGAAAGCGC")
    n = 20
    assert abs(r["f5"][n] - r["f5"][n + 1]) < 1e-10
    # duplex with no complementary pair: logZ stays at the sentinel, posterior zero
    d = oracle.duplex("AAAA", "AAAA")
    assert d["logZ2"][0] < NEG / 2 and d["post"].max() == 0.0


def test_oracle_row_sums_are_probabilities(oracle, golden):
    seq = str(golden["mc/OxyS/seq"])
    n = len(seq)
    post = oracle.inference(seq)["post"]
    P = np.zeros((n + 1, n + 1))
    for i in range(1, n + 1):
        P[i, i + 1:n + 1] = post[tri_offset(n, i) + i + 1: tri_offset(n, i) + n + 1]
    P = P + P.T
    assert P.sum(axis=1).max() <= 1.0 + 1e-9
    up = oracle.up_float(n, post.astype(np.float32))
    assert np.allclose(up, np.maximum(0, 1 - P.sum(axis=1)[1:]), atol=2e-6)


def test_algorithmic_bytes_counter(oracle):
    """B_alg = 8 B x table loads+stores of the recurrences; the FM2 term
    must match its closed form C(n+1,3) inside (2 loads) and outside (4 loads + 2 stores)."""
    n = 60
    seq = "".join("ACGU"[(7 * k * k + k) % 4] for k in range(n))
    loads, stores = oracle.count(oracle.inference, seq)
    k3 = (n + 1) * n * (n - 1) // 6
    assert loads > 6 * k3 and stores > 2 * k3
    assert loads < 6 * k3 + 3 * 496 * tri_size(n) + 10 * n * n
