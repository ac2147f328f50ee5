"""RIP tables: the text format RactIP's hidden `--rip FILE` option imports instead of running any DP
(`/root/reference/src/ractip.cpp:461-514`, `load_from_rip`).  Writing the engine's matrices in this format lets a STOCK
RactIP binary consume them unchanged:

    python -m ractip_amd.pipeline --write-rip pair.rip a.fa b.fa
    ractip --rip pair.rip --min-w 0 a.fa b.fa          # accessibility must be off: the format has no `up` table

Format: a line `Table R:` / `Table S:` / `Table I:` switches the table, every following line starting with a digit is
`i j p`.  R: bp1(i,j), 1-based, i < j.  S: sequence 2 is REVERSED in the file -- the reader stores the entry at
bp2(L2-j+1, L2-i+1).  I: hp(i, L2-j+1).  Any other line ends the current table.  Entries not listed stay 0.
"""
import numpy as np


def _tri(n, i):
    return i * (2 * (n + 1) - i - 1) // 2


def write_rip(path, s1, s2, bp1, bp2, hp, threshold=0.0):
    """Entries with p > threshold are written (p in full double precision; the reader narrows to float)."""
    n1, n2 = len(s1), len(s2)
    bp1, bp2, hp = np.asarray(bp1), np.asarray(bp2), np.asarray(hp).reshape(n1 + 1, n2 + 1)
    with open(path, "w") as f:
        f.write("Table R:\n")
        for i in range(1, n1 + 1):
            row = bp1[_tri(n1, i):_tri(n1, i) + n1 + 1]
            for j in np.flatnonzero(row[i + 1:] > threshold) + i + 1:
                f.write("%d %d %.17g\n" % (i, j, row[j]))
        f.write("\nTable S:\n")
        for a in range(1, n2 + 1):          # pair (a, b) of s2 appears as (i, j) = (n2-b+1, n2-a+1)
            row = bp2[_tri(n2, a):_tri(n2, a) + n2 + 1]
            for b in np.flatnonzero(row[a + 1:] > threshold) + a + 1:
                f.write("%d %d %.17g\n" % (n2 - b + 1, n2 - a + 1, row[b]))
        f.write("\nTable I:\n")
        for i in range(1, n1 + 1):
            for c in np.flatnonzero(hp[i, 1:] > threshold) + 1:
                f.write("%d %d %.17g\n" % (i, n2 - c + 1, hp[i, c]))
        f.write("\n")


def read_rip(path, s1, s2):
    """load_from_rip restated: (bp1, bp2, hp) as float32 in the reference's layouts."""
    n1, n2 = len(s1), len(s2)
    bp1 = np.zeros((n1 + 1) * (n1 + 2) // 2, np.float32)
    bp2 = np.zeros((n2 + 1) * (n2 + 2) // 2, np.float32)
    hp = np.zeros((n1 + 1, n2 + 1), np.float32)
    st = None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("Table R:"):
            st = "R"
        elif line.startswith("Table S:"):
            st = "S"
        elif line.startswith("Table I:"):
            st = "I"
        elif st and line[:1].isdigit():
            a, b, p = line.split()[:3]
            i, j, p = int(a), int(b), float(p)
            if st == "R":
                bp1[_tri(n1, i) + j] = p
            elif st == "S":
                bp2[_tri(n2, n2 - j + 1) + n2 - i + 1] = p
            else:
                hp[i][n2 - j + 1] = p
        else:
            st = None
    return bp1, bp2, hp
