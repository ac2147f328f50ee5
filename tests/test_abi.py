"""CPU suite: the C-ABI library builds, loads and exports every symbol that
include/ractip_hot.h declares; creation fails loudly when there is no GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ractip_hot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for want in ("rh_create", "rh_destroy", "rh_last_error", "rh_bpp", "rh_unpaired", "rh_duplex",
                 "rh_batch_upload", "rh_batch_compute", "rh_batch_results", "rh_batch_candidates"):
        assert want in syms


def test_library_exports_every_declared_symbol(hotlib):
    for name in declared_symbols():
        assert hasattr(hotlib, name), "libractip_hot.so does not export " + name


def test_library_exports_nothing_the_header_does_not_declare(hotlib):
    """The converse: every rh_* function the product library exports is part of the declared C ABI (no stray debug or tuning
    entry points in the shipped build; tuning builds -DRH_STAMPS add theirs and are not shipped)."""
    import subprocess
    import ractip_amd.hot as hot
    out = subprocess.check_output(["nm", "-D", "--defined-only", hot.LIB_PATH], text=True)
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TtWw" and ln.split()[-1].startswith("rh_")})
    assert exported, "nm found no rh_* symbols"
    extra = [e for e in exported if e not in declared_symbols()]
    assert not extra, "exported but not declared in include/ractip_hot.h: %r" % extra


def test_python_binding_lists_the_same_symbols(hotlib):
    import ractip_amd.hot as hot
    assert sorted(hot.EXPORTS) == declared_symbols()


def test_no_cpu_fallback_without_gpu(hotlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ractip_amd
    with pytest.raises(ractip_amd.RhError) as e:
        ractip_amd.Context(device=0)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under ractip_amd/ or include/ may mention it."""
    bad = []
    for base in ("ractip_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".c")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"cf_oracle|libref_contrafold|oracle/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
