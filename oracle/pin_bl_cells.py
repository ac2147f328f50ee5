#!/usr/bin/env python3
"""Independent pin of the BL* table index conventions (TEST TOOLING, build container only).

oracle/dump_bl_params.py ships the integer initialisers of /root/reference/src/boltzmann_param.c in their flat order, and both
loaders of that file (ractip_amd/csrc/vienna_loader.cpp and oracle/vienna_oracle.c) map the flat order onto (pair type,
pair type, letters...) with the conventions of the reference's copy_* loops (:5908-5971) -- written by the same hand, so a
shared misreading would be invisible.  This script does NOT use the flat order: it reads the per-block COMMENTS of the
reference's tables ("/* CG..GU */", "/* GC.A..UA */", "/* AU.CG..CG */": closing pair, loop letters, enclosed pair) and labels
every number with the indices its comment and its row / column position give it.  The result is committed as data,
tests/golden/bl_star_cells.json: [table, i, j, k, l, m, n, value] (unused indices -1), all of stack and int11, every 3rd cell of int21 /
int22, and tests/test_bl_cells.py checks the product's loaded tables against it.
Pair types CG GC GU UG AU UA @ = 1..7, letters @ A C G U = 0..4 (the row/column comments of the file, :22, :112)."""
import json
import os
import re

SRC = "/root/reference/src/boltzmann_param.c"
PT = {"CG": 1, "GC": 2, "GU": 3, "UG": 4, "AU": 5, "UA": 6, "@": 7}
NT = {"@": 0, "A": 1, "C": 2, "G": 3, "U": 4}
MAC = {"DEF": -50, "NST": 0, "INF": 1000000}
text = open(SRC).read()


def body(name):
    m = re.search(r"static\s+int\s+%s\s*\[\s*\]\s*=\s*\{(.*?)\};" % re.escape(name), text, flags=re.S)
    assert m, name
    return m.group(1)


def blocks(name):
    """[(label, [numbers])] for every '/* label */' block of the initialiser"""
    parts = re.split(r"/\*(.*?)\*/", body(name))
    out = []
    for k in range(1, len(parts), 2):
        nums = [MAC[t] if t in MAC else int(t) for t in re.findall(r"-?\d+|[A-Z]+", parts[k + 1])]
        out.append((parts[k].strip(), nums))
    return out


cells = []
# stack: one header comment with the column order, rows in the same order
hdr, nums = blocks("stack37a")[0]
order = hdr.split()
assert order == ["CG", "GC", "GU", "UG", "AU", "UA", "@"] and len(nums) == 49
for r, a in enumerate(order):
    for c, b in enumerate(order):
        cells.append(["stack", PT[a], PT[b], -1, -1, -1, -1, nums[r * 7 + c]])
for label, nums in blocks("int11_37a"):        # "CG..GU": closing pair, enclosed pair; 5x5 block over the two loop letters
    m = re.fullmatch(r"(\S+)\.\.\s*(\S+)", label)
    assert m and len(nums) == 25, label
    for p in range(25):
        cells.append(["int11", PT[m.group(1)], PT[m.group(2)], p // 5, p % 5, -1, -1, nums[p]])
for label, nums in blocks("int21_37a"):        # "CG.A..GU": + the first letter; 5x5 block over the other two
    m = re.fullmatch(r"(\S+)\.(\S)\.\.\s*(\S+)", label)
    assert m and len(nums) == 25, label
    for p in range(1, 25, 3):
        cells.append(["int21", PT[m.group(1)], PT[m.group(3)], NT[m.group(2)], p // 5, p % 5, -1, nums[p]])
for label, nums in blocks("int22_37a"):        # "CG.AC..GU": + the first two letters (A C G U only); 4x4 block over the other two
    m = re.fullmatch(r"(\S+)\.(\S)(\S)\.\.\s*(\S+)", label)
    assert m and len(nums) == 16, label
    for p in range(0, 16, 3):
        cells.append(["int22", PT[m.group(1)], PT[m.group(4)], NT[m.group(2)], NT[m.group(3)], 1 + p // 4, 1 + p % 4, nums[p]])
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bl_star_cells.json")
json.dump({"note": "BL* cells labelled by the block comments of the reference's tables (oracle/pin_bl_cells.py); "
                   "[table, i, j, k, l, m, n, value in 10 cal/mol], unused indices -1", "cells": cells}, open(dst, "w"), separators=(",", ":"))
print(len(cells), "cells ->", os.path.normpath(dst), os.path.getsize(dst), "bytes")
