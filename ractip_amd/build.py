"""Build the in-tree native libraries (hipcc, gfx950 only).

  python -m ractip_amd.build            # libractip_hot.so (+ host adapters)

hipcc cross-compiles without a GPU; the built .so files stay in-tree (git-ignored)
so that they travel with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOT_SOURCES = ["rh_api.hip", "mccaskill.hip", "mccaskill_lin.hip", "mccaskill_far.hip", "mccaskill_strip.hip", "mccaskill_small.hip", "duplex.hip", "duplex_lin.hip", "duplex_vienna.hip", "duplex_vlin.hip", "mccaskill_vienna.hip", "mccaskill_vlin.hip", "param_loader.cpp", "vienna_loader.cpp"]
HOT_LIB = os.path.join(PKG, "libractip_hot.so")
LAST_BUILD = {"compiled": [], "reused": [], "linked": False}   # what the last build_hot() did (reported by __graft_entry__.build)


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ROCm toolchain is required to build ractip_amd")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hot(force=False, verbose=True):
    """Compile every source to an object in parallel (the kernel files are template-heavy), then link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in HOT_SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "ractip_hot.h"))
    if not force and not _stale(HOT_LIB, srcs + hdrs):
        LAST_BUILD.update(compiled=[], reused=[os.path.basename(x) for x in srcs], linked=False)
        return HOT_LIB
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]

    compiled, reused = [], []

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
            if verbose:
                print("[ractip_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            compiled.append(os.path.basename(src))
        else:
            reused.append(os.path.basename(src))
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HOT_LIB] + objs + ["-ldl"]
    if verbose:
        print("[ractip_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    LAST_BUILD.update(compiled=sorted(compiled), reused=sorted(reused), linked=True)
    return HOT_LIB


def build_all(force=False, verbose=True):
    out = [build_hot(force, verbose)]
    host = os.path.join(PKG, "host", "Makefile")
    if os.path.exists(host):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(host)])
    return out


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
