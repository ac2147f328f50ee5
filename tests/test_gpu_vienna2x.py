"""GPU suite: the Vienna model under RH_VIENNA_SEM_20 (ViennaRNA-2.x E_IntLoop / E_ExtLoop / E_MLstem / E_Hairpin, the
HAVE_VIENNA20 branch of /root/reference/src/pf_duplex.c:128-206) with tables read from a ViennaRNA parameter file.
PARITY UNPINNED (ViennaRNA is absent; the reference holds no output of this path): the HIP kernels are compared with
oracle/vienna2x.py -- pf_duplex restated loop for loop from pf_duplex.c, and brute-force enumeration of every duplex / every
secondary structure under the published 2.x energy functions -- on synthetic tables in which every entry is distinct."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vienna2x as v2  # noqa: E402

pytestmark = pytest.mark.gpu
REL = 1e-9   # floating point: same energy model, different order of log-additions


def tri_offset(n, i):
    return i * (2 * (n + 1) - i - 1) // 2


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    T = v2.random_tables(23)
    path = str(tmp_path_factory.mktemp("par") / "synthetic_v20.par")
    v2.write_par_v20(path, T)
    return T, path


@pytest.fixture(scope="module")
def ctx20(hotlib, synth):
    import ractip_amd
    c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL, param_file=synth[1], vienna=dict(use_bl_param=False))
    yield c
    c.close()


def rand_seq(rng, n):
    return "".join("ACGU"[k] for k in rng.integers(0, 4, n))


def test_a_v20_file_selects_the_2x_semantics_and_the_log_space_kernels(ctx20):
    from ractip_amd.hot import RhError
    assert ctx20.vienna_semantics() == 2
    with pytest.raises(RhError, match="log-space"):
        ctx20.set_mode(2)
    ctx20.set_mode(0)
    ctx20.fold("GGGAAAUCCC")
    assert ctx20.last_path() == 2


def test_pf_duplex_2x_against_the_loop_for_loop_restatement(ctx20, synth):
    T, _ = synth
    rng = np.random.default_rng(7)
    pairs = [(rand_seq(rng, a), rand_seq(rng, b)) for a, b in ((12, 9), (25, 31), (40, 38), (1, 7), (33, 2))]
    # long loops of every class: a few pairable letters far apart (1xn, 2x3, long bulges, generic asymmetric loops)
    pairs.append(("GGAAAAAAAAAAAAAAAAAAAAAAAAAGCAAAGG", "CCUUUGUUC"))
    pairs.append(("GCAG", "CAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAGUUUGC"))
    pairs.append(("GANNCUG", "CAGNUC"))          # unknown letters: code 0 rows of the tables, never a pair
    for s1, s2 in pairs:
        hp, logz = ctx20.duplex(s1, s2)
        efw, ebk, pr = v2.pf_duplex(T, s1, s2)
        if efw == -math.inf:
            assert logz < -1e18 and not hp.any(), (s1, s2)
            continue
        assert logz == pytest.approx(efw, rel=REL) and efw == pytest.approx(ebk, rel=1e-10), (s1, s2)
        assert np.allclose(hp, pr, rtol=1e-8, atol=1e-12), (s1, s2, np.abs(hp - pr).max())
    assert ctx20.last_hybrid_path() == 2


def test_pf_duplex_2x_against_enumeration_of_all_duplexes(ctx20, synth):
    T, _ = synth
    rng = np.random.default_rng(9)
    for _ in range(6):
        s1, s2 = rand_seq(rng, int(rng.integers(3, 8))), rand_seq(rng, int(rng.integers(3, 8)))
        lz, prb = v2.brute_duplex(T, s1, s2)
        hp, logz = ctx20.duplex(s1, s2)
        if lz == -math.inf:
            assert logz < -1e18
            continue
        assert logz == pytest.approx(lz, rel=REL), (s1, s2)
        assert np.allclose(hp, prb, rtol=1e-8, atol=1e-12), (s1, s2)


def test_pf_fold_2x_against_enumeration_of_all_structures(ctx20, synth):
    T, _ = synth
    rng = np.random.default_rng(13)
    seqs = [rand_seq(rng, n) for n in (8, 10, 11, 12, 13)]
    seqs += ["GGGGACCCC"[0:0] + "GGGGGACUCC",      # tetraloop GGGGAC of the synthetic table inside a stem
             "ACAACGUAGC",                         # triloop CAACG
             "GACAGUACUC",                         # hexaloop ACAGUACU
             "GGACUUCGGUCAAGCC",                   # 16 nt: multiloops exist
             "GCGAAGCGAAGCGC"]
    ctx20.set_max_w(4)
    for s in seqs:
        lz, bp, up = v2.brute_fold(T, s, max_w=4)
        gbp, gup, glz = ctx20.fold(s)
        n = len(s)
        assert glz == pytest.approx(lz, rel=REL, abs=1e-9), s
        want = np.zeros_like(gbp)
        for (i, j), p in bp.items():
            want[tri_offset(n, i) + j] = p
        assert np.allclose(gbp, want, rtol=1e-8, atol=1e-12), (s, np.abs(gbp - want).max())
        assert np.allclose(gup, up, rtol=1e-8, atol=1e-11), (s, np.abs(gup - up).max())
    ctx20.set_max_w(15)


def test_two_molecule_ensemble_2x_against_enumeration(ctx20, synth):
    """co_pf_fold (the default RactIP::rnaduplex branch, src/ractip.cpp:400-458) under the 2.x stems: the loop around the
    missing backbone gap scores its stems with mismatch_exterior / single dangles"""
    T, _ = synth
    rng = np.random.default_rng(17)
    cases = [(rand_seq(rng, a), rand_seq(rng, b)) for a, b in ((5, 6), (7, 5), (4, 8), (6, 7))] + [("GGGAC", "GUCCC"), ("GGCGAAAGCC", "GGC")]
    for s1, s2 in cases:
        lz, hpb = v2.brute_cofold(T, s1, s2)
        hp, logz = ctx20.cofold(s1, s2)
        assert logz == pytest.approx(lz, rel=REL, abs=1e-9), (s1, s2)
        assert np.allclose(hp, hpb, rtol=1e-8, atol=1e-12), (s1, s2, np.abs(hp - hpb).max())


def test_batched_form_under_2x_equals_the_single_calls(ctx20):
    rng = np.random.default_rng(21)
    pairs = [(rand_seq(rng, 30), rand_seq(rng, 26)), (rand_seq(rng, 18), rand_seq(rng, 35)), (rand_seq(rng, 22), rand_seq(rng, 22))]
    ctx20.batch_upload(pairs)
    ctx20.batch_compute()
    res = [ctx20.batch_results(p) for p in range(len(pairs))]   # (the single-problem calls below reuse the context's batch)
    for r, (s1, s2) in zip(res, pairs):
        hp, lz = ctx20.duplex(s1, s2)
        bp1, up1, z1 = ctx20.fold(s1)
        assert np.allclose(r["hp"], hp, rtol=1e-10, atol=1e-14) and r["logZ"][2] == pytest.approx(lz, rel=1e-12)
        assert np.allclose(r["bp1"], bp1, rtol=1e-10, atol=1e-14) and r["logZ"][0] == pytest.approx(z1, rel=1e-12)
        assert np.allclose(r["up1"], up1, rtol=1e-10, atol=1e-13)


def test_bl_tables_under_forced_2x_semantics_run_with_zero_filled_slots(hotlib):
    """what a user without RNAlib's built-in tables gets: BL* + zeros for mismatch_exterior / _multi / _interior_1n / _23"""
    import ractip_amd
    M = ractip_amd.hot.RH_MODEL_VIENNA_BL
    c20 = ractip_amd.Context(device=0, model=M, vienna=dict(semantics=2))
    c18 = ractip_amd.Context(device=0, model=M)
    try:
        assert c20.vienna_semantics() == 2 and c18.vienna_semantics() == 1
        s1, s2 = "GGGAAAUCCCGAGCGAAAGCUC", "GAGCUUUCGCUCGGGAUUUCCC"
        h20, z20 = c20.duplex(s1, s2)
        h18, z18 = c18.duplex(s1, s2)
        assert np.isfinite(z20) and np.isfinite(z18) and abs(z20 - z18) > 1e-6          # the ends and the 1xn loops differ
        assert h20.min() >= 0.0 and h20.max() <= 1.0 + 1e-12
        b20, u20, f20 = c20.fold(s1)
        assert np.isfinite(f20) and b20.min() >= 0.0 and b20.max() <= 1.0 + 1e-12
    finally:
        c20.close()
        c18.close()
