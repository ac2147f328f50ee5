"""CPU suite: the product's binding of the BL* tables (ractip_amd/csrc/vienna_loader.cpp) against cells that were labelled
INDEPENDENTLY of the flat array order -- from the block comments of the reference's own tables
(/root/reference/src/boltzmann_param.c:136-5830, "/* CG.A..GU */" = closing pair, loop letter(s), enclosed pair) by
oracle/pin_bl_cells.py, committed as tests/golden/bl_star_cells.json.  The CPU restatement (oracle/vienna_oracle.c) reads
the same flat file with its own loader; here it is the PRODUCT loader that is pinned.  No GPU is needed: the hook runs the
host-side loader only."""
import ctypes
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = {"stack": 0, "int11": 1, "int21": 2, "int22": 3}


def test_loaded_bl_tables_match_cells_labelled_by_the_reference_comments(hotlib):
    lib = ctypes.CDLL(os.path.join(ROOT, "ractip_amd", "libractip_hot.so"))
    lib.rh_debug_vienna_cell.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_double)]
    lib.rh_debug_vienna_cell.restype = ctypes.c_int
    path = os.path.join(ROOT, "ractip_amd", "data", "vienna_bl_star.params").encode()
    cells = json.load(open(os.path.join(ROOT, "tests", "golden", "bl_star_cells.json")))["cells"]
    assert len(cells) > 3000 and {c[0] for c in cells} == set(TABLE)
    e = ctypes.c_double()
    bad = []
    for tab, i, j, k, l, m, n, val in cells:
        assert lib.rh_debug_vienna_cell(path, TABLE[tab], i, j, k, l, m, n, ctypes.byref(e)) == 0, (tab, i, j, k, l, m, n)
        if abs(e.value - val) > 1e-6:
            bad.append((tab, i, j, k, l, m, n, val, e.value))
    assert not bad, bad[:5]
    # the labels discriminate: reading int21 with its first and last loop letter exchanged, or with the two pairs exchanged, must
    # disagree with the comment-labelled cells in many places (int11 alone is almost symmetric under such swaps)
    i21 = [c for c in cells if c[0] == "int21"]
    swapped_letters = swapped_pairs = 0
    for tab, i, j, k, l, m, n, val in i21:
        lib.rh_debug_vienna_cell(path, 2, i, j, m, l, k, n, ctypes.byref(e))
        swapped_letters += abs(e.value - val) > 1e-6
        lib.rh_debug_vienna_cell(path, 2, j, i, k, l, m, n, ctypes.byref(e))
        swapped_pairs += abs(e.value - val) > 1e-6
    assert swapped_letters > 200 and swapped_pairs > 0, (swapped_letters, swapped_pairs)   # (the BL* int21 is nearly pair-symmetric)
