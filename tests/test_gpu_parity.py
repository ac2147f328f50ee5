"""GPU suite (-m gpu): the HIP path, called through the C ABI, against
  (1) the committed golden vectors from the reference's own engines,
  (2) the CPU oracle on seeded random inputs,
  (3) size-independent properties at BASELINE.json's full sizes.
Tolerances (BASELINE.json north_star): bp/hp/up within 1e-6 relative (double);
log-partition values within 1e-9 absolute (log units)."""
import numpy as np
import pytest

from _oracle import NEG, assert_log_close, assert_prob_close, tri_offset, tri_size
from ractip_amd.seqgen import random_pair, random_pairs

pytestmark = pytest.mark.gpu

REL = 1e-6


def rnd(rng, n):
    return "".join(rng.choice(list("ACGU"), n))


def test_bpp_vs_golden_all_bundled_and_edge_sequences(ctx, golden):
    for nm in golden["mc_names"]:
        seq = str(golden["mc/%s/seq" % nm])
        bp, z = ctx.bpp(seq)
        assert abs(z - float(golden["mc/%s/logZ" % nm])) < 1e-9, nm
        assert_prob_close(bp, golden["mc/%s/post" % nm], rel=REL, what="bp " + nm)


def test_duplex_vs_golden(ctx, golden):
    for key in golden["dx_names"]:
        a, b = key.split("+")
        s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
        hp, z = ctx.duplex(s1, s2)
        assert_log_close([z], golden["dx/%s/logZ2" % key][:1], tol=1e-9, what="logZ " + key)
        assert_prob_close(hp, golden["dx/%s/post" % key], rel=REL, what="hp " + key)


def test_config2_copa_copt_all_matrices(ctx, golden, oracle):
    """BASELINE.json config 2: CopA vs CopT, bp/hp/up checked to 1e-6 vs the CPU path."""
    s1, s2 = str(golden["mc/CopA/seq"]), str(golden["mc/CopT/seq"])
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    r = ctx.batch_results(0)
    assert_prob_close(r["bp1"], golden["mc/CopA/post"], rel=REL, what="bp1")
    assert_prob_close(r["bp2"], golden["mc/CopT/post"], rel=REL, what="bp2")
    assert_prob_close(r["hp"], golden["dx/CopA+CopT/post"], rel=REL, what="hp")
    for s, up, bp in ((s1, r["up1"], r["bp1"]), (s2, r["up2"], r["bp2"])):
        ref = oracle.up_float(len(s), bp.astype(np.float32))  # ractip.cpp:213-222 in float
        assert np.abs(up.astype(np.float32) - ref).max() < 2e-6
    assert abs(r["logZ"][0] - float(golden["mc/CopA/logZ"])) < 1e-9
    assert abs(r["logZ"][1] - float(golden["mc/CopT/logZ"])) < 1e-9
    assert abs(r["logZ"][2] - golden["dx/CopA+CopT/logZ2"][0]) < 1e-9


def test_n500_vs_golden(ctx, golden):
    """BASELINE.json config 3 inputs (mt19937(12345) pair, n=500)."""
    s1, s2 = random_pair(500)
    assert s1 == str(golden["mc500/mt500a/seq"]) and s2 == str(golden["mc500/mt500b/seq"])
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    r = ctx.batch_results(0)
    for tag, bp, z in (("mt500a", r["bp1"], r["logZ"][0]), ("mt500b", r["bp2"], r["logZ"][1])):
        assert abs(z - float(golden["mc500/%s/logZ" % tag])) < 1e-8
        assert abs(bp.sum() - float(golden["mc500/%s/post_sum" % tag])) < 1e-6
        assert_prob_close(bp[golden["mc500/%s/idx" % tag]], golden["mc500/%s/val" % tag], rel=REL, what=tag)
    assert abs(r["logZ"][2] - golden["dx500/logZ2"][0]) < 1e-8
    hp = r["hp"].ravel()
    assert abs(hp.sum() - float(golden["dx500/post_sum"])) < 1e-6
    assert_prob_close(hp[golden["dx500/idx"]], golden["dx500/val"], rel=REL, what="hp n=500")


def test_n1000_vs_golden(ctx, golden):
    """n=1000: every tile row of the blocked kernels has many far blocks; reference posterior (sparse) + logZ."""
    seq = str(golden["mc1000/seq"])
    bp, z = ctx.bpp(seq)
    assert abs(z - float(golden["mc1000/logZ"])) < 1e-8       # 123.307843
    assert abs(bp.sum() - float(golden["mc1000/post_sum"])) < 1e-6
    assert_prob_close(bp[golden["mc1000/idx"]], golden["mc1000/val"], rel=REL, what="bp n=1000")


def test_random_vs_oracle_ragged_batch(ctx, oracle):
    """Seeded random pairs of unequal lengths in ONE batch (ragged), every matrix vs the oracle."""
    rng = np.random.RandomState(4242)
    lens = [(1, 1), (2, 5), (3, 3), (7, 4), (12, 33), (33, 12), (64, 65), (90, 31), (150, 97), (31, 200)]
    pairs = [(rnd(rng, a), rnd(rng, b)) for a, b in lens]
    ctx.batch_upload(pairs)
    ctx.batch_compute()
    for p, (s1, s2) in enumerate(pairs):
        r = ctx.batch_results(p)
        o1, o2, od = oracle.inference(s1), oracle.inference(s2), oracle.duplex(s1, s2)
        what = "pair %d (%d,%d)" % (p, len(s1), len(s2))
        assert abs(r["logZ"][0] - o1["logZ"]) < 1e-9 and abs(r["logZ"][1] - o2["logZ"]) < 1e-9, what
        assert_log_close([r["logZ"][2]], [od["logZ2"][0]], tol=1e-9, what=what)
        assert_prob_close(r["bp1"], o1["post"], rel=REL, what="bp1 " + what)
        assert_prob_close(r["bp2"], o2["post"], rel=REL, what="bp2 " + what)
        assert_prob_close(r["hp"], od["post"], rel=REL, what="hp " + what)


def test_unknown_letters_and_unpairable_inputs(ctx, oracle):
    for seq in ("AAAAAAAAAA", "GGGGTTTTNNNNCCCC", "acguACGUnnnnGGGAAACCC", "G", "GC"):
        bp, z = ctx.bpp(seq)
        o = oracle.inference(seq)
        assert abs(z - o["logZ"]) < 1e-10
        assert_prob_close(bp, o["post"], rel=REL, what=seq)
    hp, z = ctx.duplex("AAAA", "AAAAAA")
    assert z < NEG / 2 and hp.max() == 0.0


def test_batch_composition_does_not_change_results(ctx):
    """Results of a problem are bit-identical whether it runs alone or inside a ragged batch.  The sweeps choose their launch
    organisation by the longest sequence they see (strips of eight diagonals from 40 letters on), so sequences shorter than that get
    a pass of their own next to longer ones (rh_api.hip: launch_mc_lin) -- the second batch below mixes 7 .. 39 letters with 300."""
    rng = np.random.RandomState(99)
    a, b = rnd(rng, 73), rnd(rng, 58)
    bp_alone, z_alone = ctx.bpp(a)
    hp_alone, zd_alone = ctx.duplex(a, b)
    ctx.batch_upload([(rnd(rng, 120), rnd(rng, 40)), (a, b), (rnd(rng, 9), rnd(rng, 130))])
    ctx.batch_compute()
    r = ctx.batch_results(1)
    assert np.array_equal(r["bp1"], bp_alone) and r["logZ"][0] == z_alone
    assert np.array_equal(r["hp"], hp_alone) and r["logZ"][2] == zd_alone
    shorts = [rnd(rng, n) for n in (7, 12, 33, 39, 40, 41)]
    alone = [ctx.bpp(s) for s in shorts]
    ctx.batch_upload([(shorts[0], rnd(rng, 300)), (shorts[1], shorts[2]), (rnd(rng, 150), shorts[3]), (shorts[4], shorts[5])])
    ctx.batch_compute()
    got = {0: ("bp1", 0, 0), 1: ("bp1", 1, 0), 2: ("bp2", 1, 1), 3: ("bp2", 2, 1), 4: ("bp1", 3, 0), 5: ("bp2", 3, 1)}
    for k, (key, p, kz) in got.items():
        r = ctx.batch_results(p)
        assert np.array_equal(r[key], alone[k][0]) and r["logZ"][kz] == alone[k][1], len(shorts[k])


def test_properties_full_size(ctx):
    """n=500 batch and one n=2000 pair (BASELINE.json configs 3 and 4): size-independent checks --
    probabilities in [0,1], every letter pairs with total probability <= 1, up = 1 - row sums,
    hp row/column sums <= 1, nothing lands on non-complementary letters."""
    def check(s1, s2, r):
        for s, bp, up in ((s1, r["bp1"], r["up1"]), (s2, r["bp2"], r["up2"])):
            n = len(s)
            assert np.isfinite(bp).all() and bp.min() >= 0.0 and bp.max() <= 1.0
            P = np.zeros((n + 1, n + 1))
            iu = np.triu_indices(n + 1, 0)
            P[iu] = bp  # reference triangular layout == row-major upper triangle incl. diagonal
            assert P[0].max() == 0.0 and np.diag(P).max() == 0.0
            rows = (P + P.T).sum(axis=1)[1:]
            assert rows.max() <= 1.0 + 1e-9
            assert np.abs(up - np.maximum(0.0, 1.0 - rows)).max() < 1e-12
            codes = np.array(["ACGU".index(c) for c in s])
            ok = np.zeros((4, 4), bool)
            for x, y in ((0, 3), (3, 0), (1, 2), (2, 1), (2, 3), (3, 2)):
                ok[x, y] = True
            bad = ~ok[codes[:, None], codes[None, :]]
            assert P[1:, 1:][bad & (P[1:, 1:] > 0)].size == 0
        hp = r["hp"]
        assert np.isfinite(hp).all() and hp.min() >= 0 and hp.max() <= 1
        assert hp[0].max() == 0 and hp[:, 0].max() == 0
        assert hp.sum(axis=1).max() <= 1 + 1e-9 and hp.sum(axis=0).max() <= 1 + 1e-9

    pairs = random_pairs(4, 500)
    ctx.batch_upload(pairs)
    ctx.batch_compute()
    for p, (s1, s2) in enumerate(pairs):
        check(s1, s2, ctx.batch_results(p))
    s1, s2 = random_pair(2000)
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    r = ctx.batch_results(0)
    check(s1, s2, r)
    assert abs(r["logZ"][0] - 258.796119) < 1e-5  # SURVEY 8c known answer, InferenceEngine<double>, n=2000


def test_n2000_vs_golden(ctx, golden):
    """BASELINE.json config 4 (n=2000/2000): reference logZ and sparse posteriors of the mt19937(12345) pair."""
    if "mc2000/logZ" not in golden.files:
        pytest.skip("n=2000 golden vectors not generated")
    s1, s2 = random_pair(2000)
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    r = ctx.batch_results(0)
    assert abs(r["logZ"][0] - float(golden["mc2000/logZ"])) < 1e-7
    assert abs(r["bp1"].sum() - float(golden["mc2000/post_sum"])) < 1e-5
    assert_prob_close(r["bp1"][golden["mc2000/idx"]], golden["mc2000/val"], rel=REL, what="bp n=2000")
    assert abs(r["logZ"][2] - golden["dx2000/logZ2"][0]) < 1e-6
    hp = r["hp"].ravel()
    assert abs(hp.sum() - float(golden["dx2000/post_sum"])) < 1e-5
    assert_prob_close(hp[golden["dx2000/idx"]], golden["dx2000/val"], rel=REL, what="hp n=2000")
    # the second sequence of the pair (oracle/gen_golden.py, GOLDEN_ONLY=2000b)
    assert abs(r["logZ"][1] - float(golden["mc2000b/logZ"])) < 1e-7
    assert abs(r["bp2"].sum() - float(golden["mc2000b/post_sum"])) < 1e-5
    assert_prob_close(r["bp2"][golden["mc2000b/idx"]], golden["mc2000b/val"], rel=REL, what="bp2 n=2000")


def test_linear_path_is_taken_and_falls_back_on_overflow(ctx, oracle):
    """Ordinary inputs run on the linear fast path; a long perfect GC helix drives the scaled
    partition function out of the double range, which must be detected and recomputed in log space."""
    if ctx.path_name != "auto":
        pytest.skip("fallback logic belongs to the auto path")
    s1, s2 = random_pair(300)
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    assert ctx.last_path() == 1
    helix = "G" * 700 + "AAAA" + "C" * 700          # log Z ~ 1.5 per nucleotide >> the scale exponent
    bp, z = ctx.bpp(helix)
    assert ctx.last_path() == 3
    o = oracle.inference(helix)
    assert abs(z - o["logZ"]) < 1e-7 * abs(o["logZ"])
    assert_prob_close(bp, o["post"], rel=REL, what="GC helix via fallback")


def test_mixed_batch_only_the_flagged_problems_fall_back(ctx, oracle):
    """One perfect GC helix among ordinary pairs: only that sequence (and the duplex of its pair) leaves the double range
    and is recomputed in log space; every other problem keeps its linear-path bits (LogSpace.hpp:232-244 has no cliff)."""
    if ctx.path_name != "auto":
        pytest.skip("fallback logic belongs to the auto path")
    pairs = random_pairs(6, 300, seed=4242)
    ctx.batch_upload(pairs)
    ctx.batch_compute()
    assert ctx.last_path() == 1 and ctx.batch_fallbacks(0) == [] and ctx.batch_fallbacks(1) == []
    base = [ctx.batch_results(p) for p in range(len(pairs))]
    helix = "G" * 348 + "AAAA" + "C" * 348                      # log Z ~ 1.5 per nucleotide >> the scale exponent: e^966 scaled
    mixed = list(pairs)
    mixed[3] = (helix, pairs[3][1])
    ctx.batch_upload(mixed)
    ctx.batch_compute()
    # WHICH mechanism: the helix is held by the third scale exponent of the linear path (1.5 per unit span), not by the log-space kernels
    assert ctx.last_path() == 3 and ctx.batch_fallbacks(2) == [6] and ctx.batch_fallbacks(0) == []
    for p in range(len(pairs)):
        r = ctx.batch_results(p)
        if p != 3:   # untouched problems: bit for bit what the all-ordinary batch gave
            for k in ("bp1", "bp2", "up1", "up2", "hp", "logZ"):
                assert np.array_equal(r[k], base[p][k]), (p, k)
        else:
            o1, od = oracle.inference(helix), oracle.duplex(helix, pairs[3][1])
            assert abs(r["logZ"][0] - o1["logZ"]) < 1e-7 * abs(o1["logZ"])
            assert_prob_close(r["bp1"], o1["post"], rel=REL, what="helix bp via per-problem fallback")
            assert np.array_equal(r["bp2"], base[3]["bp2"])         # its partner stayed on the linear path
            assert abs(r["logZ"][2] - od["logZ2"][0]) < 1e-7 * abs(od["logZ2"][0])
            assert_prob_close(r["hp"], od["post"], rel=REL, what="helix duplex")
    # a duplex that overflows while its sequences do not: a random GC 600-mer and its reverse complement (log Z of the duplex ~ 3 per pair)
    rng = np.random.RandomState(17)
    a = "".join(rng.choice(list("GC"), 600))
    b = "".join({"G": "C", "C": "G"}[ch] for ch in reversed(a))
    ctx.batch_upload(pairs[:3] + [(a, b)] + pairs[3:])
    ctx.batch_compute()
    # WHICH mechanism: the duplex scale-exponent ladder (another exponent on the linear duplex kernels), not the log-space kernels
    assert ctx.batch_fallbacks(3) == [3] and ctx.batch_fallbacks(1) == [] and ctx.last_hybrid_path() == 3
    od = oracle.duplex(a, b)
    assert abs(ctx.batch_results(3)["logZ"][2] - od["logZ2"][0]) < 1e-9 * abs(od["logZ2"][0])
    assert_prob_close(ctx.batch_results(3)["hp"], od["post"], rel=REL, what="overflowing duplex via the per-pair ladder")
    assert np.array_equal(ctx.batch_results(0)["hp"], base[0]["hp"])


def test_scale_exponent_ladder(hotlib, monkeypatch):
    """Sequences that leave the double range with the default scale exponent are recomputed on the LINEAR kernels with another
    exponent before the log-space kernels are tried (rh_batch_fallbacks which = 2).  Three 1100-nt chains of stable hairpins (log Z
    0.58 per nucleotide: 1e220 scaled with s = 0.12, in range with s = 0.45) among ordinary sequences: they are rescaled, nothing goes
    to log space, every other problem keeps its bits, and the rescaled results equal those of the log-space path (RH_SCALE_LADDER=0)."""
    import ractip_amd
    rng = np.random.default_rng(5)
    comp = {"G": "C", "C": "G"}

    def hairpins(n):
        s = ""
        while len(s) < n:
            stem = "".join(rng.choice(list("GC"), size=10))
            s += stem + "AAAA" + "".join(comp[ch] for ch in reversed(stem)) + "AA"
        return s[:n]
    rnd = lambda n: "".join(rng.choice(list("ACGU"), size=n))
    seqs = [rnd(1100), hairpins(1100), rnd(640), hairpins(1100), hairpins(1100), rnd(1100), rnd(300), rnd(900)]
    pairs = list(zip(seqs[0::2], seqs[1::2]))

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ractip_amd.Context(device=0)
        try:
            c.batch_upload(pairs)
            c.batch_compute()
            return c.last_path(), c.batch_fallbacks(0), c.batch_fallbacks(2), [c.batch_results(p) for p in range(len(pairs))]
        finally:
            c.close()
            for k in env:
                monkeypatch.delenv(k)

    path, logged, rescaled, res = run({})
    assert path == 3 and logged == [] and rescaled == [1, 3, 4]
    path0, logged0, rescaled0, ref = run({"RH_SCALE_LADDER": "0"})
    assert path0 == 3 and logged0 == [1, 3, 4] and rescaled0 == []
    for p, (r, r0) in enumerate(zip(res, ref)):
        for which, key, up in ((0, "bp1", "up1"), (1, "bp2", "up2")):
            if 2 * p + which in (1, 3, 4):
                assert abs(r["logZ"][which] - r0["logZ"][which]) < 1e-9 * abs(r0["logZ"][which])
                assert_prob_close(r[key], r0[key], rel=REL, what="rescaled sequence %d" % (2 * p + which))
                assert np.abs(r[up] - r0[up]).max() < 1e-9
            else:
                assert np.array_equal(r[key], r0[key]) and np.array_equal(r[up], r0[up]) and r["logZ"][which] == r0["logZ"][which]


def test_duplex_scale_exponent_ladder(hotlib, monkeypatch):
    """Pairs of long complementary strands (log Z of the duplex ensemble 1.2 - 1.6 per unit of i + (L2+1-j), against the 0.65 the default
    exponent assumes) leave the double range on the linear duplex kernels: they are recomputed there with another exponent
    (rh_batch_fallbacks which = 3), NOT by the log-space kernels; results equal the log-space path's to 1e-9 and every other pair of the
    batch keeps its bits."""
    import ractip_amd
    rng = np.random.default_rng(11)
    comp = {"A": "U", "U": "A", "G": "C", "C": "G"}
    rnd_ = lambda n, al="ACGU": "".join(rng.choice(list(al), size=n))
    def compl(n):
        a = rnd_(n, "GC")
        return a, "".join(comp[ch] for ch in reversed(a))
    pairs = [(rnd_(300), rnd_(280)), compl(600), (rnd_(120), rnd_(400)), compl(450), (rnd_(64), rnd_(64)), (rnd_(200), rnd_(210)), compl(700), (rnd_(90), rnd_(33))]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ractip_amd.Context(device=0)
        try:
            c.batch_upload(pairs); c.batch_compute()
            return c.last_path(), c.batch_fallbacks(1), c.batch_fallbacks(3), [c.batch_results(p) for p in range(len(pairs))], c.last_hybrid_path()
        finally:
            c.close()
            for k in env:
                monkeypatch.delenv(k)

    _, logd0, resc0, ref, _ = run({"RH_SCALE_LADDER": "0"})
    assert resc0 == [] and set(logd0) >= {1, 6} and set(logd0) <= {1, 3, 6}     # (the 450-mer pair stays inside the range on some exponent-free margin)
    _, logd, resc, res, hpath = run({})
    assert resc == logd0 and logd == [] and hpath == 3
    for p, (r, r0) in enumerate(zip(res, ref)):
        if p in logd0:
            assert abs(r["logZ"][2] - r0["logZ"][2]) < 1e-9 * abs(r0["logZ"][2]) and r0["logZ"][2] > 400
            assert_prob_close(r["hp"], r0["hp"], rel=REL, what="hp of the rescaled pair %d" % p)
        else:
            assert np.array_equal(r["hp"], r0["hp"]) and r["logZ"][2] == r0["logZ"][2]
        assert np.array_equal(r["bp1"], r0["bp1"])


def test_scale_exponent_is_remembered(hotlib):
    """A batch most of whose sequences needed another exponent moves the context there: the next batch of the kind runs on it
    at once (path 1); a later batch of ordinary long sequences that vanish under that exponent moves it back."""
    import ractip_amd
    rng = np.random.default_rng(6)
    comp = {"G": "C", "C": "G"}

    def hairpins(n):
        s = ""
        while len(s) < n:
            stem = "".join(rng.choice(list("GC"), size=10))
            s += stem + "AAAA" + "".join(comp[ch] for ch in reversed(stem)) + "AA"
        return s[:n]
    rnd = lambda n: "".join(rng.choice(list("ACGU"), size=n))
    structured = [(hairpins(1100), hairpins(1000)) for _ in range(4)]
    ordinary = [(rnd(2000), rnd(1900)) for _ in range(4)]
    c = ractip_amd.Context(device=0)
    try:
        # default: no memory -- the second batch pays the failed pass again and returns the same bits (history-free results)
        c.batch_upload(structured); c.batch_compute()
        assert c.last_path() == 3 and c.batch_fallbacks(2) == list(range(8)) and c.batch_fallbacks(0) == []
        first = [c.batch_results(p) for p in range(4)]
        c.batch_upload(structured); c.batch_compute()
        assert c.last_path() == 3 and c.batch_fallbacks(2) == list(range(8))
        for r, r0 in zip([c.batch_results(p) for p in range(4)], first):
            assert np.array_equal(r["bp1"], r0["bp1"]) and np.array_equal(r["logZ"], r0["logZ"])
        c.set_scale_memory(True)
        c.batch_upload(structured); c.batch_compute()
        assert c.last_path() == 3
        c.batch_upload(structured); c.batch_compute()
        assert c.last_path() == 1 and c.batch_fallbacks(2) == []          # straight on the exponent that worked
        for r, r0 in zip([c.batch_results(p) for p in range(4)], first):
            assert np.allclose(r["logZ"], r0["logZ"], rtol=1e-12, atol=0)
            assert_prob_close(r["bp1"], r0["bp1"], rel=1e-9, what="second batch on the remembered exponent")
        c.batch_upload(ordinary); c.batch_compute()                           # 0.12 per nucleotide under s = 0.45: 1e-280 scaled
        assert c.last_path() == 3 and c.batch_fallbacks(2) == list(range(8)) and c.batch_fallbacks(0) == []
        moved = [c.batch_results(p) for p in range(4)]
        c.batch_upload(ordinary); c.batch_compute()
        assert c.last_path() == 1                                            # back on the default exponent
        fresh = ractip_amd.Context(device=0)
        try:
            fresh.batch_upload(ordinary); fresh.batch_compute()
            assert fresh.last_path() == 1
            for p in range(4):
                r, r0, r1 = c.batch_results(p), fresh.batch_results(p), moved[p]
                assert np.array_equal(r["bp1"], r0["bp1"]) and np.array_equal(r["logZ"], r0["logZ"])   # the same kernels, the same exponent
                assert_prob_close(r1["bp2"], r0["bp2"], rel=1e-9, what="ordinary sequence through the ladder")
        finally:
            fresh.close()
    finally:
        c.close()


def test_real_and_gc_rich_sequences_stay_on_the_linear_path(ctx, golden):
    """The scale exponent is tuned on random ACGU (log Z per nucleotide 0.11-0.13); the bundled RNAs sit at 0.13-0.23 and a
    70 % GC sequence higher still: all of them must stay inside the double range on the fast path, up to n = 2000."""
    if ctx.path_name != "auto":
        pytest.skip("records the path the auto mode takes")
    names = [str(n) for n in golden["mc_names"] if len(str(golden["mc/%s/seq" % n])) >= 30]
    seqs = [str(golden["mc/%s/seq" % n]) for n in names]
    ctx.batch_upload([(s, s) for s in seqs])
    ctx.batch_compute()
    assert ctx.last_path() == 1, ("bundled sequences fell back", [names[k // 2] for k in ctx.batch_fallbacks(0)])
    rng = np.random.default_rng(7)
    gc = "".join(rng.choice(list("GCAU"), 2000, p=[0.35, 0.35, 0.15, 0.15]))
    ctx.batch_upload([(gc, gc[::-1])])
    ctx.batch_compute()
    z = ctx.batch_results(0)["logZ"]
    assert ctx.last_path() == 1, "70 %% GC, n = 2000: log Z per nucleotide %.3f left the double range" % (z[0] / 2000)
    assert 0.15 < z[0] / 2000 < 0.45


def test_zscore_batch_of_1000_shuffles_sampled_against_the_oracle(ctx, oracle):
    """BASELINE config 5, DP stage: the 1000 dinucleotide shuffles of OxyS / fhlA (--zscore=12 --seed=1, src/ractip.cpp:1636-1643)
    in one device pass; 16 of them checked in full (bp1, bp2, hp, three log Z) against the CPU oracle."""
    import os
    from ractip_amd import shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fa = [l.strip() for l in open(os.path.join(root, "ractip_amd", "data", "config5_OxyS_fhlA.fa")) if not l.startswith(">")]
    pairs = shard.zscore_shuffles(fa[0], fa[1], 12, 1000, 1)
    ctx.batch_upload(pairs)
    ctx.batch_compute()
    logz = ctx.batch_logz()
    assert logz.shape == (1000, 3) and np.all(np.isfinite(logz))
    for p in list(range(0, 1000, 67)) + [999]:
        s1, s2 = pairs[p]
        r = ctx.batch_results(p)
        o1, o2, od = oracle.inference(s1), oracle.inference(s2), oracle.duplex(s1, s2)
        assert abs(r["logZ"][0] - o1["logZ"]) < 1e-9 and abs(r["logZ"][1] - o2["logZ"]) < 1e-9
        assert abs(r["logZ"][2] - od["logZ2"][0]) < 1e-9 and np.array_equal(r["logZ"], logz[p])
        assert_prob_close(r["bp1"], o1["post"], rel=REL, what="shuffle %d bp1" % p)
        assert_prob_close(r["bp2"], o2["post"], rel=REL, what="shuffle %d bp2" % p)
        assert_prob_close(r["hp"], od["post"], rel=REL, what="shuffle %d hp" % p)


def test_threshold_candidates_match_reference_scans(ctx, oracle, golden):
    """rh_batch_candidates == the reference's scans (ractip.cpp:557-568, 578-589, 598-608) applied to the
    float-narrowed oracle matrices: same entries, same (row-major) order, p > threshold in float."""
    s1, s2 = str(golden["mc/OxyS/seq"]), str(golden["mc/fhlA/seq"])
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    r = ctx.batch_results(0)
    th_ss, th_hy, th_ac = np.float32(0.5), np.float32(0.1), np.float32(0.003)   # cmdline.c defaults
    for which, s, dense in ((0, s1, r["bp1"]), (1, s2, r["bp2"])):
        n = len(s)
        want = [(i, j) for i in range(1, n + 1) for j in range(i + 1, n + 1)
                if np.float32(dense[tri_offset(n, i) + j]) > th_ss]
        got = ctx.batch_candidates(0, which, float(th_ss))
        assert [(i, j) for i, j, _ in got] == want
        assert all(np.float32(p) == np.float32(dense[tri_offset(n, i) + j]) for i, j, p in got)
        o = oracle.inference(s)["post"]
        assert want == [(i, j) for i in range(1, n + 1) for j in range(i + 1, n + 1) if np.float32(o[tri_offset(n, i) + j]) > th_ss]
    want = [(i, j) for i in range(1, len(s1) + 1) for j in range(1, len(s2) + 1) if np.float32(r["hp"][i, j]) > th_hy]
    got = ctx.batch_candidates(0, 2, float(th_hy))
    assert [(i, j) for i, j, _ in got] == want and len(want) > 0
    got = ctx.batch_candidates(0, 3, float(th_ac))
    assert [i for i, _, _ in got] == [i for i in range(len(s1)) if np.float32(r["up1"][i]) > th_ac]
    assert len(ctx.batch_candidates(0, 2, float(th_hy), cap=5)) == 5      # cap truncates the copy, not the count


def test_bulk_results_and_batched_candidates_equal_per_pair_calls(ctx):
    rng = np.random.RandomState(31)
    pairs = [(rnd(rng, a), rnd(rng, b)) for a, b in ((40, 70), (111, 90), (64, 64), (5, 130))]
    ctx.batch_upload(pairs)
    ctx.batch_compute()
    bulk = ctx.batch_results_all()
    for p in range(len(pairs)):
        one = ctx.batch_results(p)
        for k in ("bp1", "bp2", "up1", "up2", "hp", "logZ"):
            assert np.array_equal(np.asarray(bulk[p][k]), np.asarray(one[k])), (p, k)
    for which, th in ((0, 0.5), (1, 0.5), (2, 0.1), (3, 0.003), (4, 0.003)):
        rec, first = ctx.batch_candidates_all(which, th)
        assert first[0] == 0 and first[-1] == len(rec)
        for p in range(len(pairs)):
            mine = [tuple(r) for r in rec[first[p]:first[p + 1]].tolist()]
            assert mine == ctx.batch_candidates(p, which, th), (which, p)


def test_custom_parameter_file(hotlib, oracle, tmp_path):
    """rh_create(param_file): a perturbed weight file must change the results exactly as it changes the oracle's."""
    import ctypes
    import ractip_amd
    from _oracle import PARAMS
    rng = np.random.RandomState(8)
    lines = [l.split() for l in open(PARAMS)]
    path = tmp_path / "perturbed.params"
    rng.shuffle(lines)                                   # binding is by name: order must not matter
    path.write_text("".join("%s %r\n" % (k, float(v) + 0.05 * rng.randn()) for k, v in lines))
    m = oracle.L.cfo_load_params(str(path).encode())
    assert m
    seq, s2 = rnd(rng, 90), rnd(rng, 60)
    n = len(seq)
    post = np.zeros(tri_size(n))
    z = oracle.L.cfo_inference(ctypes.c_void_p(m), seq.encode(), n, post.ctypes.data, None, None)
    c = ractip_amd.Context(device=0, param_file=str(path))
    bp, zg = c.bpp(seq)
    assert abs(zg - z) < 1e-9
    assert_prob_close(bp, post, rel=REL, what="perturbed weights")
    S = (n + 1) * (len(s2) + 1)
    hp_ref, z2 = np.zeros(S), np.zeros(2)
    oracle.L.cfo_duplex(ctypes.c_void_p(m), seq.encode(), n, s2.encode(), len(s2), hp_ref.ctypes.data, None, None, z2.ctypes.data)
    hp, zd = c.duplex(seq, s2)
    assert abs(zd - z2[0]) < 1e-9
    assert_prob_close(hp, hp_ref, rel=REL, what="perturbed duplex")
    c.close()
    assert abs(z - oracle.inference(seq)["logZ"]) > 1e-3   # the perturbation really changed the model


def test_vienna_bl_duplex_vs_its_cpu_restatement(hotlib, golden):
    """RH_MODEL_VIENNA_BL (pf_duplex with BL* energies, ViennaRNA-1.8 semantics): PARITY UNPINNED against the
    reference; the HIP kernels are checked against oracle/vienna_oracle.c, which is itself checked against
    brute-force enumeration (tests/test_vienna_oracle.py)."""
    import ractip_amd
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
    rng = np.random.RandomState(77)
    cases = [(str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])) for a, b in
             (("DIS", "DIS"), ("CopA", "CopT"), ("OxyS", "fhlA"), ("Tar", "Tarstar"), ("R1inv", "R2inv"))]
    cases += [(rnd(rng, a), rnd(rng, b)) for a, b in ((1, 1), (3, 9), (40, 33), (120, 95), (64, 200))]
    cases += [("GGGTTTNNNCCC", "GGGAAACCCUUU")]
    for s1, s2 in cases:
        hp, z = c.duplex(s1, s2)
        o = vo.pf_duplex(s1, s2)
        if not np.isfinite(o["logZ"]):
            assert z < NEG / 2 and hp.max() == 0
            continue
        assert abs(z - o["logZ"]) < 1e-9 * max(1.0, abs(z)), (len(s1), len(s2))
        assert_prob_close(hp, o["pr"], rel=REL, what="vienna duplex %d/%d" % (len(s1), len(s2)))
    c.close()


@pytest.fixture(scope="module", params=["auto", "log"])
def vctx(hotlib, request):
    """Vienna-BL context per arithmetic path: "auto" = scaled linear kernels (mccaskill_vlin.hip) with log-space
    fallback, "log" = log-space kernels only (mccaskill_vienna.hip)."""
    import ractip_amd
    c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
    c.set_mode({"auto": 0, "log": 1}[request.param])
    c.path_name = request.param
    yield c
    c.close()


def test_vienna_bl_mccaskill_and_accessibility_vs_cpu_restatement(vctx, golden):
    """RH_MODEL_VIENNA_BL rnafold path (pf_fold bp + pf_unstru up, src/ractip.cpp:248-382): PARITY UNPINNED against
    the reference (ViennaRNA absent); HIP kernels == oracle/vienna_oracle.c, which == brute-force enumeration of all
    structures (tests/test_vienna_oracle.py)."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    rng = np.random.RandomState(91)
    seqs = [str(golden["mc/%s/seq" % a]) for a in ("DIS", "CopA", "CopT", "OxyS", "fhlA", "Tar")]
    seqs += ["GGGAAACCCAGGGAAACCCA", "GGAAACCCAGGGAAACCC", "GGCGAAAGCCAAGGCGAAAGCCAA", "GGGTTTNNNCCCAAAGGG", "A", "GC", "GAAAC", "GGAAACC"]
    seqs += [rnd(rng, n) for n in (6, 17, 63, 64, 65, 130, 200)]
    for s in seqs:
        n = len(s)
        o = vo.mccaskill(s, max_w=15)
        bp, up, z = vctx.fold(s)
        assert vctx.max_w == 15 and up.shape == (n, 15)
        assert abs(z - o["logZ"]) < 1e-9 * max(1.0, abs(z)), n
        assert_prob_close(bp, o["post"], rel=REL, what="vienna bp n=%d" % n)
        assert_prob_close(up, o["up"], rel=REL, abs_floor=1e-11, what="vienna up n=%d" % n)
        bp2, z2 = vctx.bpp(s)
        assert np.array_equal(bp, bp2) and z == z2
    assert vctx.last_path() == (1 if vctx.path_name == "auto" else 2)
    # another width through rh_unpaired
    s = seqs[1]
    assert_prob_close(vctx.unpaired(s, max_w=4), vo.mccaskill(s, max_w=4)["up"], rel=REL, abs_floor=1e-11, what="up w=4")
    vctx.set_max_w(15)


def test_vienna_bl_structure_constraints(vctx, golden):
    """use_constraint_ (src/ractip.cpp:271-291): pf_fold under fold_constrained.  The engine's allowed-pair mask (C++) and the
    test's (numpy) are independent restatements of ViennaRNA-1.8 make_ptypes; the CPU restatement under the mask is itself
    checked against brute force below."""
    import ractip_amd
    from _oracle import ViennaOracle, ractip_constraint
    vo = ViennaOracle()
    s = "GGGAAACCCAGGGAAACCCA"
    for c in ("....................", "xxx.................", "(......)............", "<<<......>>>........", ".(......)......x....",
              "((....))..(((...))).", ".|||....x...........", "(((...)))", ""):
        o = vo.mccaskill(s, max_w=15, constraint=c)
        b = vo.fold_bruteforce(s, max_w=15, constraint=c)
        assert abs(o["logZ"] - b["logZ"]) < 1e-11 and np.abs(o["post"] - b["post"]).max() < 1e-11 and np.abs(o["up"] - b["up"]).max() < 1e-11
        bp, up, z = vctx.fold(s, constraint=c)
        assert abs(z - o["logZ"]) < 1e-9, c
        assert_prob_close(bp, o["post"], rel=REL, what="constrained bp " + c)
        assert_prob_close(up, o["up"], rel=REL, abs_floor=1e-11, what="constrained up " + c)
    # a bundled RNA with a RactIP-style structure line: brackets of the interaction and 'e' become 'x'
    seq = str(golden["mc/OxyS/seq"])
    n = len(seq)
    free = vo.mccaskill(seq)["post"]
    # force a moderately likely pair (i,j) far from the block 43..52 that the interaction brackets / 'e' turn into 'x'
    cand = [(free[tri_offset(n, i) + j], i, j) for i in range(1, 40) for j in range(i + 4, 41) if 0.05 < free[tri_offset(n, i) + j] < 0.6]
    _, fi, fj = max(cand)
    line = list("." * n)
    line[fi - 1], line[fj - 1] = "(", ")"
    line[42:47] = "[[[[["
    line[47:52] = "eeeee"
    line[60:62] = "<<"
    cons = ractip_constraint("".join(line), n)
    assert cons[42:52] == "x" * 10 and cons[fi - 1] == "("
    o = vo.mccaskill(seq, max_w=15, constraint=cons)
    bp, up, z = vctx.fold(seq, constraint=cons)
    assert abs(z - o["logZ"]) < 1e-9 * abs(z) and abs(z - vo.mccaskill(seq)["logZ"]) > 1e-3
    assert_prob_close(bp, o["post"], rel=REL, what="OxyS constrained bp")
    assert_prob_close(up, o["up"], rel=REL, abs_floor=1e-11, what="OxyS constrained up")
    assert up[42:52, 0].min() > 1 - 1e-12                       # 'x' letters are unpaired
    assert bp[tri_offset(n, fi) + fi + 1:tri_offset(n, fi) + n + 1].sum() == bp[tri_offset(n, fi) + fj]   # fi pairs with fj or nothing
    assert np.array_equal(vctx.bpp(seq)[0], vctx.fold(seq)[0])                                  # the mask does not linger
    for bad in ("(((", ")", "(..)"):   # unbalanced; forced pair G-G
        with pytest.raises(ractip_amd.RhError):
            vctx.bpp("GGGAAACCC" if bad != "(..)" else "GAAG", constraint=bad)


def test_vienna_bl_constrained_two_molecule_ensemble(vctx, golden):
    """co_pf_fold under the constraint string RactIP builds with use_constraint_ (src/ractip.cpp:405-447): forced
    intermolecular pairs '(' ')' across the cut, 'x' for letters paired inside their own molecule."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    s1, s2 = "GGGAAACCCAGG", "CCUGGGAAACCC"
    for c in ("........................", "(......................)", "((....................))", "xxx.....................",
              "......(((......)))......"):
        o, b = vo.cofold(s1, s2, constraint=c), vo.cofold(s1, s2, bruteforce=True, constraint=c)
        assert abs(o["logZ"] - b["logZ"]) < 1e-11 and np.abs(o["post"] - b["post"]).max() < 1e-11
        hp, z = vctx.cofold(s1, s2, constraint=c)
        assert abs(z - o["logZ"]) < 1e-9, c
        assert_prob_close(hp, o["hp"], rel=REL, what="constrained cofold " + c)
    a, b = str(golden["mc/CopA/seq"]), str(golden["mc/CopT/seq"])
    n1, n2 = len(a), len(b)
    c = list("." * (n1 + n2))
    c[20], c[n1 + n2 - 21] = "(", ")"          # CopA[21] with CopT[n2-20]: complementary antisense letters
    c[0:5] = "xxxxx"
    c = "".join(c)
    o = vo.cofold(a, b, constraint=c)
    hp, z = vctx.cofold(a, b, constraint=c)
    assert abs(z - o["logZ"]) < 1e-9 * abs(z)
    assert_prob_close(hp, o["hp"], rel=REL, what="CopA/CopT constrained")
    assert hp[1:6].max() == 0 and abs(hp[21, n2 - 20] - hp[21].sum()) < 1e-15
    assert np.array_equal(vctx.cofold(a, b)[0], vctx.cofold(a, b, constraint="." * (n1 + n2))[0])


def test_vienna_bl_full_size_pair(vctx):
    """BASELINE config 3 under the default-CLI model: one n=500/500 pair (mt19937(12345) stream), rnafold x2 with
    accessibility at width 15 + the two-molecule ensemble over N=1000 (several block-product tiles, the cut in the middle of
    a 64-cell group), against the CPU restatement; plus size-independent properties."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    s1, s2 = random_pair(500)
    vctx.set_hybrid(True)
    try:
        vctx.batch_upload([(s1, s2)])
        vctx.batch_compute()
        r = vctx.batch_results(0)
    finally:
        vctx.set_hybrid(False)
    o1, oc = vo.mccaskill(s1, max_w=15), vo.cofold(s1, s2)
    assert abs(r["logZ"][0] - o1["logZ"]) < 1e-9 * abs(o1["logZ"]) and abs(r["logZ"][2] - oc["logZ"]) < 1e-9 * abs(oc["logZ"])
    assert_prob_close(r["bp1"], o1["post"], rel=REL, what="bp1 n=500")
    assert_prob_close(r["up1"], o1["up"], rel=REL, abs_floor=1e-11, what="up1 n=500")
    assert_prob_close(r["hp"], oc["hp"], rel=REL, what="hp n=500/500")
    # every letter of s2 is unpaired or paired exactly once (width-1 accessibility + pair probabilities)
    n = len(s2)
    paired = np.zeros(n + 1)
    for i in range(1, n + 1):
        row = r["bp2"][tri_offset(n, i):tri_offset(n, i) + n + 1]
        paired[i] += row[i + 1:].sum()
        paired[i + 1:] += row[i + 1:]
    assert np.abs(paired[1:] + r["up2"][:, 0] - 1).max() < 1e-9
    assert (np.diff(r["up2"], axis=1) <= 1e-12).all()


def test_vienna_bl_accessibility_organisations_agree(hotlib, monkeypatch):
    """The gap-probability sums behind `up` have two organisations: one thread per letter and gap length (vlin_acc_gaps, all 30 lengths;
    RH_ACC_WIDE=0) and the default (lengths 1-2 chunked over the inner spans + vlin_acc_gaps_wide: one lane per run of eight
    (own gap, other gap) pairs, outer values from an LDS ring).  Same sums in a different order: equal to 1e-12, ragged lengths
    incl. sequences shorter than the ring (56 spans) and than one block of inner spans."""
    import ractip_amd
    rng = np.random.default_rng(77)
    seqs = ["".join(rng.choice(list("ACGU"), size=n)) for n in (9, 23, 57, 64, 131, 300, 412, 38)]
    pairs = list(zip(seqs[0::2], seqs[1::2]))

    def run(wide):
        monkeypatch.setenv("RH_ACC_WIDE", wide)
        c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
        try:
            c.set_max_w(15)
            c.batch_upload(pairs)
            c.batch_compute()
            assert c.last_path() == 1   # scaled linear path
            return [c.batch_results(p) for p in range(len(pairs))]
        finally:
            c.close()
            monkeypatch.delenv("RH_ACC_WIDE")

    a, b = run("0"), run("1")
    for (s1, s2), x, y in zip(pairs, a, b):
        for k, s in (("up1", s1), ("up2", s2)):
            assert x[k].shape == y[k].shape and x[k].shape[0] == len(s)
            assert np.abs(x[k] - y[k]).max() <= 1e-12, (k, len(s))
    # the last step, E + I + multiloop streams per (letter, width): one thread per letter and width (RH_ACC_FINAL_T=0) or one thread
    # per letter with the operands of its fifteen widths staged in LDS -- the same terms in the same order: the same bits
    monkeypatch.setenv("RH_ACC_FINAL_T", "0")
    try:
        c0 = run("1")
    finally:
        monkeypatch.delenv("RH_ACC_FINAL_T")
    for x, y in zip(c0, b):
        assert np.array_equal(x["up1"], y["up1"]) and np.array_equal(x["up2"], y["up2"])


def test_vienna_bl_two_molecule_organisations_agree(hotlib, monkeypatch):
    """The scaled linear two-molecule sweeps have three switches that must not change hp or log Z beyond the order of a few sums:
    RH_CO_SEED (1: one-strand inside cells copied from the single folds, the staged rows masked by the inner cell's strands and the
    filters unrolled; 0: every cell computed, per-lane filter limits), RH_CO_WINDOW (1: only the groups around the cut are launched),
    and both together.  Ragged pairs: cuts at a group boundary (64), next to one (63, 65), a one-letter strand, s1 longer than s2."""
    import ractip_amd
    from _oracle import ViennaOracle
    rng = np.random.default_rng(78)
    rs = lambda n: "".join(rng.choice(list("ACGU"), size=n))
    pairs = [(rs(64), rs(70)), (rs(63), rs(130)), (rs(65), rs(41)), (rs(1), rs(90)), (rs(200), rs(33)), (rs(129), rs(127))]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
        try:
            c.set_hybrid(True)
            c.batch_upload(pairs)
            c.batch_compute()
            assert c.last_path() == 1   # scaled linear path
            return [c.batch_results(p) for p in range(len(pairs))]
        finally:
            c.close()
            for k in env:
                monkeypatch.delenv(k)

    base = run({})
    vo = ViennaOracle()
    for (s1, s2), r in list(zip(pairs, base))[:3]:
        o = vo.cofold(s1, s2)
        assert abs(r["logZ"][2] - o["logZ"]) < 1e-8
        assert_prob_close(r["hp"], o["hp"], rel=REL, what="cofold hp %d/%d" % (len(s1), len(s2)))
    for env in ({"RH_CO_SEED": "0"}, {"RH_CO_WINDOW": "0"}, {"RH_CO_SEED": "0", "RH_CO_WINDOW": "0"}):
        got = run(env)
        for (s1, s2), r, r0 in zip(pairs, got, base):
            assert abs(r["logZ"][2] - r0["logZ"][2]) < 1e-10, (env, len(s1), len(s2))
            assert_prob_close(r["hp"], r0["hp"], rel=1e-10, what="%r hp %d/%d" % (env, len(s1), len(s2)))


def test_vienna_bl_two_molecule_graph_is_rekeyed_when_the_shortest_cut_changes(hotlib):
    """Round-2 advisor finding: the captured launch graph of the two-molecule sweeps bakes in the window of groups around the cut,
    which follows from the SHORTEST s1 of the batch.  Two batches with the same count and maxima but another shortest s1 must
    not replay the first batch's window: hp / log Z of the second batch equal those of a fresh context."""
    import ractip_amd
    rng = np.random.default_rng(79)
    rs = lambda n: "".join(rng.choice(list("ACGU"), size=n))
    first = [(rs(200), rs(200)), (rs(200), rs(200))]
    second = [(rs(200), rs(200)), (rs(70), rs(200))]

    def fresh(pairs):
        c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
        try:
            c.set_hybrid(True)
            c.batch_upload(pairs)
            c.batch_compute()
            assert c.last_path() == 1
            return [c.batch_results(p) for p in range(len(pairs))]
        finally:
            c.close()

    want = fresh(second)
    c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
    try:
        c.set_hybrid(True)
        for pairs in (first, second):
            c.batch_upload(pairs)
            c.batch_compute()
            assert c.last_path() == 1
        got = [c.batch_results(p) for p in range(len(second))]
    finally:
        c.close()
    for r, r0 in zip(got, want):
        assert abs(r["logZ"][2] - r0["logZ"][2]) < 1e-10
        assert_prob_close(r["hp"], r0["hp"], rel=1e-10, what="hp after a batch with another shortest cut")
        assert r["hp"].max() > 1e-3   # the short-cut pair's cross-strand cells were really computed


def test_vienna_bl_scale_exponent_ladder(hotlib, monkeypatch):
    """Vienna-BL model: a batch with sequences outside the double range of the default exponent (chains of stable hairpins, log Z
    0.87 per nucleotide at 900 nt) is run again on the linear kernels with another exponent -- whole batch: folds, accessibility and
    the two-molecule sweeps -- instead of the log-space kernels; results equal those of the log-space path (RH_SCALE_LADDER=0), and
    the next batch of the kind starts on the exponent that worked."""
    import ractip_amd
    rng = np.random.default_rng(8)
    comp = {"G": "C", "C": "G"}

    def hairpins(n):
        s = ""
        while len(s) < n:
            stem = "".join(rng.choice(list("GC"), size=10))
            s += stem + "AAAA" + "".join(comp[ch] for ch in reversed(stem)) + "AA"
        return s[:n]
    rnd = lambda n: "".join(rng.choice(list("ACGU"), size=n))
    pairs = [(hairpins(900), rnd(300)), (rnd(420), hairpins(800)), (rnd(350), rnd(500)), (hairpins(850), hairpins(700))]

    def run(env, twice=False):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
        try:
            c.set_hybrid(True)
            c.set_scale_memory(twice)
            c.batch_upload(pairs); c.batch_compute()
            out = (c.last_path(), c.batch_fallbacks(2), [c.batch_results(p) for p in range(len(pairs))])
            if twice:
                c.batch_upload(pairs); c.batch_compute()
                out += (c.last_path(), c.batch_fallbacks(2))
            return out
        finally:
            c.close()
            for k in env:
                monkeypatch.delenv(k)

    path, rescaled, res, path2, rescaled2 = run({}, twice=True)
    assert path == 3 and rescaled == [0, 3, 6]                 # (the 700-nt chain stays inside the range: 1e179 scaled)
    assert path2 == 1 and rescaled2 == []                     # second batch: straight on the exponent that worked
    path0, rescaled0, ref = run({"RH_SCALE_LADDER": "0"})
    assert path0 == 3 and rescaled0 == []
    for p, (r, r0) in enumerate(zip(res, ref)):
        assert np.allclose(r["logZ"], r0["logZ"], rtol=1e-9, atol=0), p
        assert_prob_close(r["bp1"], r0["bp1"], rel=REL, what="bp1 of pair %d" % p)
        assert_prob_close(r["bp2"], r0["bp2"], rel=REL, what="bp2 of pair %d" % p)
        assert_prob_close(r["hp"], r0["hp"], rel=REL, what="hp of pair %d" % p)
        assert_prob_close(r["up1"], r0["up1"], rel=REL, abs_floor=1e-11, what="up1 of pair %d" % p)


def test_vienna_bl_flagged_pairs_are_recomputed_alone(hotlib, monkeypatch):
    """Vienna-BL, per-pair route of the ladder: ONE chain of stable hairpins among seven ordinary pairs.  Only the pair that holds it is
    recomputed (on a helper context: another exponent, or log space), every other pair keeps the bits of a batch without the chain, and
    the chain's results equal the whole-batch route's (RH_PAIR_HELPER=0) to 1e-9."""
    import ractip_amd
    rng = np.random.default_rng(9)
    comp = {"G": "C", "C": "G"}

    def hairpins(n):
        s = ""
        while len(s) < n:
            stem = "".join(rng.choice(list("GC"), size=10))
            s += stem + "AAAA" + "".join(comp[ch] for ch in reversed(stem)) + "AA"
        return s[:n]
    rnd_ = lambda n: "".join(rng.choice(list("ACGU"), size=n))
    plain = [(rnd_(300), rnd_(260)), (rnd_(150), rnd_(333)), (rnd_(64), rnd_(65)), (rnd_(400), rnd_(90)),
             (rnd_(220), rnd_(210)), (rnd_(900), rnd_(190)), (rnd_(128), rnd_(127)), (rnd_(77), rnd_(300))]   # (the same longest length in both batches)
    mixed = list(plain)
    mixed[5] = (hairpins(900), plain[5][1])

    def run(pairs, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
        try:
            c.set_hybrid(True)
            c.batch_upload(pairs); c.batch_compute()
            return c.last_path(), c.batch_fallbacks(2), c.batch_fallbacks(0), [c.batch_results(p) for p in range(len(pairs))]
        finally:
            c.close()
            for k in env:
                monkeypatch.delenv(k)

    path0, _, _, base = run(plain, {})
    assert path0 == 1
    path, resc, logd, res = run(mixed, {"RH_PAIR_HELPER": "2"})       # (2: the helper whenever at most half of the pairs are flagged; default: one in 16)
    assert path == 3 and sorted(resc + logd) == [10, 11]            # the one pair, whichever mechanism held it on the helper
    _, _, _, whole = run(mixed, {"RH_PAIR_HELPER": "0"})
    for p in range(len(plain)):
        if p != 5:
            for k in ("bp1", "bp2", "up1", "up2", "hp", "logZ"):
                assert np.array_equal(res[p][k], base[p][k]), (p, k)
        else:
            assert np.allclose(res[p]["logZ"], whole[p]["logZ"], rtol=1e-9, atol=0)
            for k in ("bp1", "bp2", "hp"):
                assert_prob_close(res[p][k], whole[p][k], rel=REL, what="%s of the recomputed pair" % k)
            assert_prob_close(res[p]["up1"], whole[p]["up1"], rel=REL, abs_floor=1e-11, what="up1 of the recomputed pair")


def test_vienna_bl_linear_path_falls_back_on_overflow(vctx):
    """A 1200-nt perfect GC helix has log Z ~ 2.5 per nucleotide: the scaled linear values leave the double range, the
    device flags it and the batch is recomputed in log space; results equal the log-space context's."""
    if vctx.path_name != "auto":
        pytest.skip("fallback logic belongs to the auto path")
    import ractip_amd
    s = "G" * 600 + "AAAA" + "C" * 600
    bp, z = vctx.bpp(s)
    assert vctx.last_path() == 3
    ref = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
    ref.set_mode(1)
    bp2, z2 = ref.bpp(s)
    ref.close()
    # (recomputed in log space -- then bit for bit the log-space context's result -- or held by another scale exponent of the linear path)
    assert np.isfinite(z) and z > 1000 and abs(z - z2) < 1e-9 * z2
    assert_prob_close(bp, bp2, rel=REL, what="GC helix after the fallback")
    n = len(s)
    o = tri_offset(n, 300)
    assert bp[o + 301:o + n + 1].sum() > 0.99              # a G in the middle of the run is paired (the helix may slip)


def test_vienna_bl_linear_duplex_path_and_its_fallback(vctx):
    """pf_duplex under the Vienna-BL model runs in scaled linear space (duplex_vlin.hip: four anti-diagonals per launch) and
    equals the log-space kernels to rounding; two perfectly complementary 700-mers (log Z ~ 1.3 per letter pair, five times
    the scale the tables are stored at) leave the double range, are flagged and recomputed in log space."""
    if vctx.path_name != "auto":
        pytest.skip("path selection belongs to the auto path")
    import ractip_amd
    ref = ractip_amd.Context(device=0, model=ractip_amd.hot.RH_MODEL_VIENNA_BL)
    ref.set_mode(1)
    rng = np.random.RandomState(17)
    try:
        vctx.set_hybrid(False)
        for n1, n2 in ((61, 64), (125, 190), (331, 280)):   # around the 62-column groups of the four-diagonal kernel
            s1, s2 = rnd(rng, n1), rnd(rng, n2)
            hp, z = vctx.duplex(s1, s2)
            assert vctx.last_hybrid_path() == 1
            hp2, z2 = ref.duplex(s1, s2)
            assert abs(z - z2) < 1e-10
            assert_prob_close(hp, hp2, rel=1e-9, what="linear vs log pf_duplex %d/%d" % (n1, n2))
        s1, s2 = "GC" * 350, "GC" * 350
        hp, z = vctx.duplex(s1, s2)
        assert vctx.last_hybrid_path() == 3
        hp2, z2 = ref.duplex(s1, s2)
        assert np.array_equal(hp, hp2) and z == z2 and np.isfinite(z) and z > 500
    finally:
        ref.close()


def test_vienna_bl_pair_batch(vctx, golden):
    """Batched form under the Vienna-BL model: everything RactIP::solve consumes on its default path with --duplex
    (bp1, bp2, up1, up2 at width 15, hp), ragged lengths, plus the on-device threshold scans of up."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    rng = np.random.RandomState(92)
    pairs = [(str(golden["mc/CopA/seq"]), str(golden["mc/CopT/seq"])), (str(golden["mc/OxyS/seq"]), str(golden["mc/fhlA/seq"])),
             (rnd(rng, 150), rnd(rng, 90)), (rnd(rng, 31), rnd(rng, 160))]
    vctx.batch_upload(pairs)
    vctx.batch_compute()
    allr = vctx.batch_results_all()
    for p, (s1, s2) in enumerate(pairs):
        r = vctx.batch_results(p)
        o1, o2, od = vo.mccaskill(s1, max_w=15), vo.mccaskill(s2, max_w=15), vo.pf_duplex(s1, s2)
        assert_prob_close(r["bp1"], o1["post"], rel=REL, what="bp1")
        assert_prob_close(r["bp2"], o2["post"], rel=REL, what="bp2")
        assert_prob_close(r["up1"], o1["up"], rel=REL, abs_floor=1e-11, what="up1")
        assert_prob_close(r["up2"], o2["up"], rel=REL, abs_floor=1e-11, what="up2")
        assert_prob_close(r["hp"], od["pr"], rel=REL, what="hp")
        assert np.abs(r["logZ"] - np.array([o1["logZ"], o2["logZ"], od["logZ"]])).max() < 1e-8
        for k in ("bp1", "bp2", "up1", "up2", "hp"):
            assert np.array_equal(np.asarray(allr[p][k]), r[k]), k
        # accessible regions (src/ractip.cpp:621-627): up[i][j] > th_ac, region (i, i+j)
        th = np.float32(0.003)
        want = [(i, j) for i in range(len(s1)) for j in range(15) if np.float32(r["up1"][i, j]) > th]
        got = [(i, j) for i, j, _ in vctx.batch_candidates(p, 3, float(th))]
        assert got == want
    recs, first = vctx.batch_candidates_all(4, 0.003)
    for p, (s1, s2) in enumerate(pairs):
        mine = recs[first[p]:first[p + 1]]
        single = vctx.batch_candidates(p, 4, 0.003)
        assert [(int(a), int(b)) for a, b in zip(mine["i"], mine["j"])] == [(i, j) for i, j, _ in single]


def test_vienna_bl_cofold_hybridization(vctx, golden):
    """RH_HYBRID_COFOLD: hp from the two-molecule ensemble (co_pf_fold semantics, the default branch of
    RactIP::rnaduplex, src/ractip.cpp:400-458).  PARITY UNPINNED; HIP == oracle/vienna_oracle.c == brute force."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    rng = np.random.RandomState(93)
    vctx.set_hybrid(True)
    try:
        cases = [(str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])) for a, b in (("DIS", "DIS"), ("CopA", "CopT"), ("OxyS", "fhlA"))]
        cases += [("GGGAAACCC", "GGGUUUCCC"), ("G", "C"), ("GGGAAACCCAGGGAAACCCA", "UGGG"), ("CCCA", "GGGAAACCCAGGGAAACCCAUGGG"),
                  ("GGGTTTNNNCCC", "GGGAAACCCUUU")]
        cases += [(rnd(rng, a), rnd(rng, b)) for a, b in ((1, 1), (3, 9), (40, 33), (64, 1), (90, 75))]
        for s1, s2 in cases:
            hp, z = vctx.duplex(s1, s2)
            o = vo.cofold(s1, s2)
            assert abs(z - o["logZ"]) < 1e-9 * max(1.0, abs(z)), (len(s1), len(s2))
            assert_prob_close(hp, o["hp"], rel=REL, what="cofold hp %d/%d" % (len(s1), len(s2)))
            assert vctx.last_hybrid_path() in ((1, 3) if vctx.path_name == "auto" else (2,))
        hp, z = vctx.duplex(*cases[-1])                   # random 90/75: stays inside the double range on the linear path
        assert vctx.last_hybrid_path() == (1 if vctx.path_name == "auto" else 2)
        # batched: bp/up of the single molecules are untouched by the choice of hp source
        pairs = cases[:3] + [(rnd(rng, 70), rnd(rng, 52))]
        vctx.batch_upload(pairs)
        vctx.batch_compute()
        for p, (s1, s2) in enumerate(pairs):
            r = vctx.batch_results(p)
            o = vo.cofold(s1, s2)
            assert_prob_close(r["hp"], o["hp"], rel=REL, what="batched cofold hp")
            assert abs(r["logZ"][2] - o["logZ"]) < 1e-8
            assert_prob_close(r["bp1"], vo.mccaskill(s1)["post"], rel=REL, what="bp1")
            assert_prob_close(r["up2"], vo.mccaskill(s2, max_w=15)["up"], rel=REL, abs_floor=1e-11, what="up2")
    finally:
        vctx.set_hybrid(False)


def test_errors_are_reported_not_swallowed(ctx):
    import ractip_amd
    with pytest.raises(ractip_amd.RhError):
        ctx.unpaired("ACGU", max_w=5)      # the CONTRAfold path has width-1 accessibility only
    with pytest.raises(ractip_amd.RhError):
        ctx.batch_upload([("ACGU", "")])


def test_sweep_organisations_agree(hotlib, oracle, monkeypatch):
    """The strips of eight diagonals with the banded near/far split (default), the fused two-diagonal sweeps with the block-aligned
    split, the look-ahead pair of launches, the one-diagonal-per-launch kernels, strips with four wavefronts per group, strips for
    one sweep only (mixed masked / plain block-diagonal-2 tiles), two-level block products forced on, the
    block products with / without packed operand tiles and the duplex sweep with four / two anti-diagonals per launch are
    the same arithmetic up to summation order: every variant against
    the CPU oracle at 1e-6 and against the default at 1e-10, on lengths around the 63/64-column group edges and the
    16-letter blocks."""
    import ractip_amd
    rng = np.random.RandomState(5)
    seqs = [rnd(rng, n) for n in (62, 63, 64, 65, 127, 128, 129, 190, 331, 40, 57, 58, 71, 115)]
    pairs = [(seqs[0], seqs[3]), (seqs[4], seqs[1]), (seqs[7], seqs[8])]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = ractip_amd.Context(device=0)
        try:
            out = [c.bpp(s) for s in seqs]
            c.batch_upload(pairs)
            c.batch_compute()
            res = [c.batch_results(p) for p in range(len(pairs))]
        finally:
            c.close()
            for k in env:
                monkeypatch.delenv(k)
        return out, res

    base, base_pairs = run({})
    for (bp, z), s in zip(base, seqs):
        o = oracle.inference(s)
        assert abs(z - o["logZ"]) < 1e-9
        assert_prob_close(bp, o["post"], rel=REL, what="default n=%d" % len(s))
    # (RH_STRIP_FILT=0: the dense single-branch filter instead of the factored one, A(t) B(|l1-l2|) + sparse residual)
    for env in ({"RH_STRIP_FILT": "0"}, {"RH_STRIP": "0"}, {"RH_STRIP": "1"}, {"RH_STRIP": "2"}, {"RH_STRIP_W": "4"}, {"RH_STRIP_XCD": "0"}, {"RH_FAR2": "1"},
                {"RH_STRIP": "0", "RH_FAR2": "1"}, {"RH_STRIP": "0", "RH_LOOKAHEAD": "1"}, {"RH_STRIP": "0", "RH_LOOKAHEAD": "0"}, {"RH_FAR_PK": "0"},
                {"RH_STRIP": "0", "RH_LOOKAHEAD": "0", "RH_LIN_W": "8"}, {"RH_STRIP": "0", "RH_LIN_W": "8"}, {"RH_DX_W": "8"}, {"RH_DX_QUAD": "0"},
                # (RH_SMALL=1: sequences of 8..109 letters by the one-workgroup-per-sequence kernel, the others of the same batch by the sweeps)
                {"RH_SMALL": "1"}):
        got, got_pairs = run(env)
        for (bp, z), (bp0, z0), s in zip(got, base, seqs):
            assert abs(z - z0) < 1e-10, (env, len(s))
            assert_prob_close(bp, bp0, rel=1e-10, what="%r n=%d" % (env, len(s)))
        for r, r0 in zip(got_pairs, base_pairs):
            assert_prob_close(r["hp"], r0["hp"], rel=1e-10, what="%r hp" % (env,))
            assert_prob_close(r["bp1"], r0["bp1"], rel=1e-10, what="%r bp1" % (env,))
            assert np.allclose(r["logZ"], r0["logZ"], rtol=0, atol=1e-10)


def test_short_sequences_one_workgroup_each(hotlib, oracle, monkeypatch):
    """RH_SMALL=1 (mccaskill_small.hip): every sequence of 8..109 letters is folded by one workgroup with its tables in LDS, whatever
    else the batch holds.  Lengths around the limits (7/8, 64/65/66 = one or two column groups, 109/110), a ragged batch with long
    sequences in it, unpairable and unknown letters: bp / up / log Z against the CPU oracle, and bit-identical alone / inside the batch."""
    import ractip_amd
    monkeypatch.setenv("RH_SMALL", "1")
    rng = np.random.RandomState(77)
    lens = [7, 8, 9, 31, 33, 53, 64, 65, 66, 67, 100, 108, 109, 110, 150]
    seqs = [rnd(rng, n) for n in lens] + ["A" * 40, "GGGGNNNNCCCCNNNNGGGGAAAACCCC", "GC" * 30]
    c = ractip_amd.Context(device=0)
    try:
        alone = [c.bpp(s) for s in seqs]
        for (bp, z), s in zip(alone, seqs):
            o = oracle.inference(s)
            assert abs(z - o["logZ"]) < 1e-9, len(s)
            assert_prob_close(bp, o["post"], rel=REL, what="short n=%d" % len(s))
        pairs = [(seqs[k], seqs[(k + 5) % len(seqs)]) for k in range(len(seqs))] + [(rnd(rng, 300), seqs[5])]
        c.batch_upload(pairs)
        c.batch_compute()
        assert c.batch_fallbacks(0) == [] and c.batch_fallbacks(2) == []
        for p, (s1, s2) in enumerate(pairs[:-1]):
            r = c.batch_results(p)
            k1, k2 = seqs.index(s1), seqs.index(s2)
            for key, kz, k, sq in (("bp1", 0, k1, s1), ("bp2", 1, k2, s2)):   # routed by its own length: the batch cannot matter
                assert np.array_equal(r[key], alone[k][0]) and r["logZ"][kz] == alone[k][1], (p, len(sq))
            ref = oracle.up_float(len(s1), r["bp1"].astype(np.float32))   # ractip.cpp:213-222 in float
            assert np.abs(r["up1"].astype(np.float32) - ref).max() < 2e-6
        r = c.batch_results(len(pairs) - 1)
        o = oracle.inference(pairs[-1][0])
        assert abs(r["logZ"][0] - o["logZ"]) < 1e-9
        assert_prob_close(r["bp1"], o["post"], rel=REL, what="the long sequence of the mixed batch")
    finally:
        c.close()
