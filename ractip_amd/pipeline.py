"""Sequences -> joint structure: the probability layer on the GPU (include/ractip_hot.h) followed by RactIP::solve's
integer programme (ractip_amd/ilp.py).  What `ractip s1.fa s2.fa` prints (/root/reference/src/ractip.cpp:1561-1610),
without z-scores and energies.

  python -m ractip_amd.pipeline [--contrafold] [--duplex] [--write-rip FILE] a.fa b.fa

--write-rip FILE: also write bp1, bp2, hp as RIP tables (ractip_amd/rip.py) for a stock `ractip --rip FILE --min-w 0`.
"""
import sys

import math

import numpy as np

from . import energy, hot, ilp, rip, shard


def read_fasta(path):
    name, seq = "", []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if seq:
                break
            name = line[1:]
        elif line:
            seq.append(line)
    return name, "".join(seq)


def probabilities(ctx, s1, s2):
    ctx.batch_upload([(s1, s2)])
    ctx.batch_compute()
    return ctx.batch_results(0)


def predict(s1, s2, model="vienna", duplex=False, device=0, options=None, ctx=None, rip_path=None):
    """model "vienna": RactIP's default path (rnafold + rnaduplex; duplex=True = --duplex, else co_pf_fold), parity
    unpinned; model "contrafold": the --contrafold path (bp from the CONTRAfold engine, width-1 up, accessibility off as
    src/ractip.cpp:1511-1517 demands), hp from the CONTRAfold duplex engine."""
    own = ctx is None
    if own:
        ctx = hot.Context(device=device, model=hot.RH_MODEL_VIENNA_BL if model == "vienna" else hot.RH_MODEL_CONTRAFOLD)
    try:
        opt = options or ilp.Options()
        if model == "vienna":
            ctx.set_max_w(max(1, opt.max_w))
            ctx.set_hybrid(not duplex)
            r = probabilities(ctx, s1, s2)
            if rip_path:
                rip.write_rip(rip_path, s1, s2, r["bp1"], r["bp2"], r["hp"])
            return ilp.solve(s1, s2, r["bp1"], r["bp2"], r["hp"], r["up1"], r["up2"], opt)
        r = probabilities(ctx, s1, s2)
        if rip_path:
            rip.write_rip(rip_path, s1, s2, r["bp1"], r["bp2"], r["hp"])
        if options is None:
            opt = ilp.Options(min_w=0)
        return ilp.solve(s1, s2, r["bp1"], r["bp2"], r["hp"], None, None, opt)
    finally:
        if own:
            ctx.close()


def _energies(s1, s2, mats, opt):
    """One iteration of what RactIP::run does with show_energy_/z-score on (src/ractip.cpp:1599-1601, 1645-1650): the joint
    programme with e1, e2, e3 and the two single-sequence programmes with e1s, e2s (all float, like the reference)."""
    r1, r2, _ = ilp.solve(s1, s2, mats["bp1"], mats["bp2"], mats["hp"], mats.get("up1"), mats.get("up2"), opt)
    f32 = np.float32
    e1, e2 = f32(energy.energy_of_structure(s1, r1, noncanonical=True)), f32(energy.energy_of_structure(s2, r2, noncanonical=True))
    e3 = f32(energy.energy_of_duplex(s1, s2, r1, r2))
    r1s, _ = ilp.solve_ss(s1, mats["bp1"], opt)
    r2s, _ = ilp.solve_ss(s2, mats["bp2"], opt)
    e1s, e2s = f32(energy.energy_of_structure(s1, r1s, noncanonical=True)), f32(energy.energy_of_structure(s2, r2s, noncanonical=True))
    return r1, r2, e1, e2, e3, e1s, e2s


def zscore(s1, s2, mode=12, num_shuffling=1000, seed=1, options=None, matrices=None, device=0):
    """The z-score loop of RactIP::run (src/ractip.cpp:1624-1670) on the default path: the native pair and `num_shuffling`
    dinucleotide shuffles (the reference's own RNG order, ractip_amd/shard.py), ALL probability matrices in one device
    pass, then the programmes and energies per iteration on the host with the reference's float accumulation order.
    `matrices(pairs) -> list of dicts` overrides the probability source (tests)."""
    opt = options or ilp.Options()
    pairs = [(s1, s2)] + shard.zscore_shuffles(s1, s2, mode, num_shuffling, seed)
    if matrices is None:
        ctx = hot.Context(device=device, model=hot.RH_MODEL_VIENNA_BL)
        try:
            ctx.set_max_w(max(1, opt.max_w))
            ctx.set_hybrid(True)
            ctx.batch_upload(pairs)
            ctx.batch_compute()
            mats = ctx.batch_results_all()
        finally:
            ctx.close()
    else:
        mats = matrices(pairs)
    f32 = np.float32
    r1, r2, e1, e2, e3, e1s, e2s = _energies(s1, s2, mats[0], opt)
    tot = f32(f32(e1 + e2) + e3)
    sm = sm2 = sms = sm2s = f32(0.0)
    for (a, b), m in zip(pairs[1:], mats[1:]):
        _, _, ee1, ee2, ee3, ee1s, ee2s = _energies(a, b, m, opt)
        ee = f32(f32(ee1 + ee2) + ee3)
        ees = f32(f32(ee - ee1s) - ee2s)
        sm, sm2 = f32(sm + ee), f32(sm2 + f32(ee * ee))          # src/ractip.cpp:1655-1656
        sms, sm2s = f32(sms + ees), f32(sm2s + f32(ees * ees))
    nsh = f32(num_shuffling)
    mean = f32(sm / nsh); var = max(f32(0.0), f32(f32(sm2 / nsh) - f32(mean * mean)))
    means = f32(sms / nsh); vars_ = max(f32(0.0), f32(f32(sm2s / nsh) - f32(means * means)))
    with np.errstate(divide="ignore", invalid="ignore"):
        z1 = f32(tot - mean) / f32(math.sqrt(var))
        z2 = f32(f32(f32(tot - e1s) - e2s) - means) / f32(math.sqrt(vars_))
    return dict(r1=r1, r2=r2, e1=float(e1), e2=float(e2), e3=float(e3), e1s=float(e1s), e2s=float(e2s), zscore=float(z1), zscore_s=float(z2))


def main(argv):
    rip_path = None
    if "--write-rip" in argv:
        k = argv.index("--write-rip")
        rip_path = argv[k + 1]
        argv = argv[:k] + argv[k + 2:]
    flags = [a for a in argv if a.startswith("--")]
    files = [a for a in argv if not a.startswith("--")]
    if len(files) != 2:
        raise SystemExit(__doc__)
    (n1, s1), (n2, s2) = read_fasta(files[0]), read_fasta(files[1])
    r1, r2, _ = predict(s1, s2, model="contrafold" if "--contrafold" in flags else "vienna", duplex="--duplex" in flags, rip_path=rip_path)
    print(">%s\n%s\n%s\n>%s\n%s\n%s" % (n1, s1, r1, n2, s2, r2))   # src/ractip.cpp:1607-1610


if __name__ == "__main__":
    main(sys.argv[1:])
