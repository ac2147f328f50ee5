#!/usr/bin/env python3
"""Extract the BL* 37-degree energy tables of the reference as DATA.

TEST/DATA TOOLING (build container only).  Reads the integer initialisers of
/root/reference/src/boltzmann_param.c:21-5868 (Andronescu et al. 2010 BL* parameters, 10 cal/mol units) and
writes them, unchanged and in the reference's own flat order, to ractip_amd/data/vienna_bl_star.params as
"name count" headers followed by the values.  How each flat array maps onto ViennaRNA's multi-dimensional
tables (the copy_* loops, boltzmann_param.c:5908-5993) is restated by the loaders that read the file.
Macros: DEF = -50, NST = 0 (boltzmann_param.c:17-18), INF = 1000000 (ViennaRNA energy_const.h).
"""
import os
import re

here = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/src/boltzmann_param.c").read()
src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
macros = {"DEF": -50, "NST": 0, "INF": 1000000}
want = ["stack37a", "mismatchH37a", "mismatchI37a", "dangle5_37a", "dangle3_37a", "int11_37a", "int21_37a", "int22_37a",
        "hairpin37a", "bulge37a", "internal_loop37a", "MLparams_a", "ninio_a"]
out = os.path.join(here, "..", "ractip_amd", "data", "vienna_bl_star.params")
with open(out, "w") as f:
    f.write("# BL* 37C energies (10 cal/mol), flat order of boltzmann_param.c; see oracle/dump_bl_params.py\n")
    for name in want:
        m = re.search(r"static\s+int\s+%s\s*\[\s*\]\s*=\s*\{(.*?)\};" % re.escape(name), src, flags=re.S)
        assert m, name
        vals = [macros[t] if t in macros else int(t) for t in re.findall(r"-?\d+|[A-Z]+", m.group(1))]
        f.write("%s %d\n" % (name[:-1] if name.endswith("a") and not name.endswith("_a") else name[:-2], len(vals)))
        for k in range(0, len(vals), 20):
            f.write(" ".join(str(v) for v in vals[k:k + 20]) + "\n")
        print(name, len(vals))
# tetraloop bonuses (boltzmann_param.c:5869-5906): closing pair + 4 loop letters, energy
tl = re.findall(r'\{\s*"([ACGU]{6})"\s*,\s*(-?\d+)\s*\}', src)
with open(out, "a") as f:
    f.write("tetraloops %d\n" % len(tl))
    for seq, e in tl:
        f.write("%s %s\n" % (seq, e))
print("tetraloops", len(tl))
print("wrote", os.path.normpath(out))
