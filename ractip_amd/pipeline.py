"""Sequences -> joint structure: the probability layer on the GPU (include/ractip_hot.h) followed by RactIP::solve's
integer programme (ractip_amd/ilp.py).  What `ractip s1.fa s2.fa` prints (/root/reference/src/ractip.cpp:1561-1610),
without z-scores and energies.

  python -m ractip_amd.pipeline [--contrafold] [--duplex] [--write-rip FILE] a.fa b.fa

--write-rip FILE: also write bp1, bp2, hp as RIP tables (ractip_amd/rip.py) for a stock `ractip --rip FILE --min-w 0`.
"""
import sys

from . import hot, ilp, rip


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
