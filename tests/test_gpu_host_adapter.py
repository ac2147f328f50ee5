"""GPU suite: the C++ mirror of RactIP's probability-layer members (ractip_amd/host)
and the pf_duplex shim, driven the way RactIP::solve drives them, against the oracle
narrowed to float as the reference's VF/VVF containers do (src/ractip.cpp:82-83)."""
import os
import subprocess

import numpy as np
import pytest

from _oracle import tri_size

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "ractip_amd", "host", "prob_cli")


@pytest.fixture(scope="module")
def cli(hotlib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "ractip_amd", "host")])
    return CLI


def run(cli, *args):
    out = subprocess.run([cli] + list(args), check=True, capture_output=True, text=True).stdout.split("\n")
    return [l for l in out if l]


def take(lines, pos, tag):
    hdr = lines[pos].split()
    assert hdr[0] == tag, (hdr, tag)
    dims = [int(x) for x in hdr[1:]]
    cnt = int(np.prod(dims))
    vals = np.array([float(x) for x in lines[pos + 1:pos + 1 + cnt]])
    return vals.reshape(dims) if len(dims) > 1 else vals, pos + 1 + cnt


def close32(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, what
    # float narrowing of two doubles that agree to 1e-6 relative: at most ~1 float ulp apart
    assert np.all(np.abs(a - b) <= 2e-6 * np.maximum(np.abs(b), 1e-6) + 1e-12), what


def test_contrafold_member(cli, oracle, golden):
    seq = str(golden["mc/RyhB/seq"])
    n = len(seq)
    lines = run(cli, "contrafold", seq)
    off, p = take(lines, 0, "offset")
    bp, p = take(lines, p, "bp")
    up, p = take(lines, p, "up")
    assert list(off.astype(int)) == [i * (2 * (n + 1) - i - 1) // 2 for i in range(n + 1)]  # ractip.cpp:257
    assert bp.size == tri_size(n) and up.size == n
    o = oracle.inference(seq)
    close32(bp, o["post"], "bp")
    assert np.abs(up.astype(np.float32) - oracle.up_float(n, o["post"].astype(np.float32))).max() < 2e-6


def test_contraduplex_threshold(cli, oracle, golden):
    s1, s2 = str(golden["mc/Tar/seq"]), str(golden["mc/Tarstar/seq"])
    ref = oracle.duplex(s1, s2)["post"]
    th = 0.1
    hp_t, _ = take(run(cli, "contraduplex", s1, s2, str(th)), 0, "hp")
    expect = np.where(ref.astype(np.float32) >= np.float32(th), ref, 0.0)  # GetPosterior(th_hy_), ractip.cpp:237
    edge = np.abs(ref - th) < 1e-6
    close32(np.where(edge, 0, hp_t), np.where(edge, 0, expect), "contraduplex hp")


def test_rnaduplex_is_the_vienna_bl_pf_duplex_on_every_entry_point(cli, golden):
    """RactIP::rnaduplex --duplex (src/ractip.cpp:390-398) is pf_duplex() with the BL* energies.  The C++ member, the
    source-compatible pf_duplex shim and the batched default path with duplex = true are three doors to the same
    numbers (parity unpinned model: checked against its CPU restatement)."""
    from _oracle import ViennaOracle
    s1, s2 = str(golden["mc/Tar/seq"]), str(golden["mc/Tarstar/seq"])
    ref = ViennaOracle().pf_duplex(s1, s2)["pr"]
    hp, _ = take(run(cli, "rnaduplex", s1, s2), 0, "hp")
    close32(hp, ref, "rnaduplex hp vs Vienna-BL pf_duplex")
    shim, _ = take(run(cli, "pfduplex", s1, s2), 1, "hp")
    close32(shim, ref, "pf_duplex shim")
    assert np.abs(hp.astype(np.float32) - shim.astype(np.float32)).max() <= 1e-7
    lines = run(cli, "solve_default_duplex", "3", s1, s2)
    pos = 1
    _, pos = take(lines, pos, "bp")
    _, pos = take(lines, pos, "up")
    hb, pos = take(lines, pos, "hp")
    assert np.abs(hb.astype(np.float32) - hp.astype(np.float32)).max() <= 1e-7


def test_vienna_2x_semantics_through_the_member_and_the_shim(cli, tmp_path):
    """the HAVE_VIENNA20 branch of pf_duplex (src/pf_duplex.c:128-206) for whoever has the tables: -P file in the v2.0 layout,
    --no-bl-param; the member and the source-compatible shim read the same environment (PARITY UNPINNED, oracle/vienna2x.py)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vienna2x as v2
    T = v2.random_tables(31)
    par = str(tmp_path / "synthetic.par")
    v2.write_par_v20(par, T)
    s1, s2 = "GGAUCGAAGGCUAGCAUCG", "CGAUGCUUGCCUAGAUCC"
    _, _, pr = v2.pf_duplex(T, s1, s2)
    env = {"RACTIP_AMD_VIENNA_PARAMS": par, "RACTIP_AMD_NO_BL_PARAM": "1"}
    hp, _ = take(run_env(cli, env, "rnaduplex", s1, s2), 0, "hp")
    close32(hp, pr, "rnaduplex under 2.x semantics")
    shim, _ = take(run_env(cli, env, "pfduplex", s1, s2), 1, "hp")
    close32(shim, pr, "pf_duplex shim under 2.x semantics")
    hp18, _ = take(run_env(cli, dict(env, RACTIP_AMD_VIENNA_SEMANTICS="1"), "rnaduplex", s1, s2), 0, "hp")
    assert np.abs(hp18 - hp).max() > 1e-4      # the same tables under the 1.8 loop energies give another matrix


def test_in_process_shard_over_a_device_list(cli, golden):
    """ProbabilityEngine over a device list (here the one GPU twice: two contexts, two host threads): contiguous blocks of
    the pairs, results in iteration order -- identical output to the single-context run (src/ractip.cpp:1636-1663)."""
    names = ["DIS", "Tar", "Tarstar", "R1inv", "R2inv"]
    args = []
    for k in range(5):
        args += [str(golden["mc/%s/seq" % names[k]]), str(golden["mc/%s/seq" % names[(k + 2) % 5]])]
    one = run(cli, "solve", *args)
    two = run_env(cli, {"RACTIP_DEVICES": "0,0"}, "solve", *args)
    three = run_env(cli, {"RACTIP_DEVICES": "0,0,0"}, "solve", *args)
    assert one == two == three and sum(l.startswith("pair") for l in one) == 5


def run_env(cli, env, *args):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([cli] + list(args), check=True, capture_output=True, text=True, env=e).stdout.split("\n")
    return [l for l in out if l]


def test_pf_duplex_shim_surface(cli, oracle, golden):
    """double pf_duplex(s1,s2); extern double** pr_duplex; void free_pf_duplex() -- src/pf_duplex.h:25-28."""
    from _oracle import ViennaOracle
    s1, s2 = str(golden["mc/R1inv/seq"]), str(golden["mc/R2inv/seq"])
    # default: BL* energies, ViennaRNA-1.8 semantics (parity unpinned; checked against our CPU restatement)
    lines = run(cli, "pfduplex", s1, s2)
    assert lines[0].startswith("logZ")
    z = float(lines[0].split()[1])
    hp, _ = take(lines, 1, "hp")
    o = ViennaOracle().pf_duplex(s1, s2)
    assert abs(z - o["logZ"]) < 1e-9 * max(1.0, abs(z))
    assert np.abs(hp - o["pr"]).max() < 1e-9
    # CONTRAfold duplex scores on request: pinned to the reference's DuplexEngine
    lines = run_env(cli, {"RACTIP_AMD_DUPLEX_MODEL": "contrafold"}, "pfduplex", s1, s2)
    z = float(lines[0].split()[1])
    hp, _ = take(lines, 1, "hp")
    o = oracle.duplex(s1, s2)
    assert abs(z - o["logZ2"][0]) < 1e-9
    assert np.abs(hp - o["post"]).max() < 1e-9


def test_solve_probabilities_batch(cli, oracle, golden):
    names = [("DIS", "DIS"), ("R1inv", "R2inv")]
    args = []
    for a, b in names:
        args += [str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])]
    lines = run(cli, "solve", *args)
    p = 0
    for a, b in names:
        s1, s2 = str(golden["mc/%s/seq" % a]), str(golden["mc/%s/seq" % b])
        z = [float(x) for x in lines[p].split()[1:]]
        p += 1
        o1, o2, od = oracle.inference(s1), oracle.inference(s2), oracle.duplex(s1, s2)
        assert abs(z[0] - o1["logZ"]) < 1e-9 and abs(z[1] - o2["logZ"]) < 1e-9 and abs(z[2] - od["logZ2"][0]) < 1e-9
        for o in (o1, o2):
            _, p = take(lines, p, "offset")
            bp, p = take(lines, p, "bp")
            _, p = take(lines, p, "up")
            close32(bp, o["post"], "bp")
        hp, p = take(lines, p, "hp")
        close32(hp, od["post"], "hp")


def test_errors_become_logic_error_exit_code(cli):
    r = subprocess.run([cli, "nosuchmode", "ACGU"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown mode" in r.stderr


def test_default_cli_path_members_rnafold_and_cofold(cli, golden):
    """RactIP's default path: rnafold (pf_fold bp + pf_unstru up, src/ractip.cpp:308-382) and the co_pf_fold branch of
    rnaduplex (:400-458) through the C++ mirror, against the float-narrowed CPU restatement (PARITY UNPINNED model)."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    s1, s2 = str(golden["mc/DIS/seq"]), str(golden["mc/Tar/seq"])
    n = len(s1)
    lines = run(cli, "rnafold", s1, "15")
    off, p = take(lines, 0, "offset")
    bp, p = take(lines, p, "bp")
    up, p = take(lines, p, "up")
    assert list(off.astype(int)) == [i * (2 * (n + 1) - i - 1) // 2 for i in range(n + 1)]
    o = vo.mccaskill(s1, max_w=15)
    close32(bp, o["post"], "rnafold bp")
    close32(up.reshape(n, 15), o["up"], "rnafold up")
    hp, _ = take(run(cli, "cofold", s1, s2), 0, "hp")
    ref = vo.cofold(s1, s2)["hp"].astype(np.float32)
    ref[ref <= np.float32(0.1)] = 0            # p > th_hy_, src/ractip.cpp:452
    near = np.abs(vo.cofold(s1, s2)["hp"] - 0.1) < 1e-6   # entries within rounding of the threshold may fall either side
    assert np.all((np.abs(hp.astype(np.float32) - ref) <= 2e-6 * np.maximum(ref, 1e-6) + 1e-12) | near)


def test_default_path_batch(cli, golden):
    """solve_probabilities_default: everything RactIP::solve needs on its default path for several pairs in one device pass."""
    from _oracle import ViennaOracle
    vo = ViennaOracle()
    pairs = [(str(golden["mc/DIS/seq"]), str(golden["mc/DIS/seq"])), (str(golden["mc/Tar/seq"]), str(golden["mc/Tarstar/seq"]))]
    lines = run(cli, "solve_default", "7", *[x for p in pairs for x in p])
    pos = 0
    for s1, s2 in pairs:
        hdr = lines[pos].split()
        assert hdr[0] == "pair"
        z = [float(v) for v in hdr[1:]]
        bp, pos = take(lines, pos + 1, "bp")
        up, pos = take(lines, pos, "up")
        hp, pos = take(lines, pos, "hp")
        o1, o2, oc = vo.mccaskill(s1, max_w=7), vo.mccaskill(s2, max_w=7), vo.cofold(s1, s2)
        assert abs(z[0] - o1["logZ"]) < 1e-8 and abs(z[2] - oc["logZ"]) < 1e-8
        close32(bp, o1["post"], "default bp1")
        close32(up.reshape(len(s2), 7), o2["up"], "default up2")
        ref = oc["hp"].astype(np.float32)
        ref[ref <= np.float32(0.1)] = 0
        near = np.abs(oc["hp"] - 0.1) < 1e-6
        assert np.all((np.abs(hp.astype(np.float32) - ref) <= 2e-6 * np.maximum(ref, 1e-6) + 1e-12) | near)
