#!/usr/bin/env python3
"""tools/build_variant.py TAG -DFOO=... [-D...]: tuning build of libractip_hot into ractip_amd/libractip_hot_TAG.so
(use with RACTIP_HOT_LIB=ractip_amd/libractip_hot_TAG.so).  Only the kernel sources that see the macro are recompiled."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ractip_amd import build as rb
tag, defs = sys.argv[1], sys.argv[2:]
objdir = os.path.join(rb.PKG, "build", "variant_" + tag)
os.makedirs(objdir, exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function"] + defs
def one(src):
    obj = os.path.join(objdir, src + ".o")
    subprocess.check_call([rb._hipcc()] + flags + ["-c", os.path.join(rb.CSRC, src), "-o", obj])
    return obj
with ThreadPoolExecutor(6) as pool:
    objs = list(pool.map(one, rb.HOT_SOURCES))
out = os.path.join(rb.PKG, "libractip_hot_%s.so" % tag)
subprocess.check_call([rb._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
print(out)
