#!/usr/bin/env python3
"""Generate tests/golden/contrafold_golden.npz from the REFERENCE engines.

TEST TOOLING.  Runs only in the build container (needs /root/reference and
oracle/_ref/libref_contrafold.so built by `make -C oracle ref`).  The fixture
holds data only: input sequences (the reference's bundled data/*.fa plus seeded
random ones) and the reference's outputs for them in double precision:
  McCaskill (InferenceEngine<double>): logZ, posterior T(n), F5i/F5o, and for a
    few short sequences the six DP tables FCi,FMi,FM1i,FCo,FMo,FM1o;
  duplex (DuplexEngine<double>): inside/outside logZ, posterior (n1+1)(n2+1),
    and for short pairs the inside/outside tables.
Also the float-engine logZ (what RactIP itself instantiates, ractip.cpp:200-201).
"""
import ctypes, glob, os, random, sys
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
from ractip_amd.seqgen import random_pair

REF = "/root/reference"
lib = ctypes.CDLL(os.path.join(here, "_ref", "libref_contrafold.so"))
lib.ref_inference.restype = ctypes.c_double
lib.ref_inference.argtypes = [ctypes.c_char_p, ctypes.c_int] + [ctypes.c_void_p] * 3
lib.ref_duplex.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int] + [ctypes.c_void_p] * 4


def golden_2000b(out):
    """sparse posterior + logZ of the SECOND sequence of the n=2000 mt19937(12345) pair (about 3 min of reference CPU time)"""
    b2000 = random_pair(2000)[1]
    post = np.zeros(2001 * 2002 // 2)
    z = lib.ref_inference(b2000.encode(), 0, post.ctypes.data, None, None)
    out["mc2000b/logZ"] = np.array(z)
    out["mc2000b/post_sum"] = np.array(post.sum())
    idx = np.flatnonzero(post > 1e-3)
    out["mc2000b/idx"] = idx.astype(np.int64)
    out["mc2000b/val"] = post[idx]
    print("mc2000b logZ=%.9f sum=%.9f nnz=%d" % (z, post.sum(), idx.size))


if os.environ.get("GOLDEN_ONLY") == "2000b":   # add this block to the existing fixture without regenerating the rest
    dst = os.path.join(here, "..", "tests", "golden", "contrafold_golden.npz")
    old = dict(np.load(dst))
    golden_2000b(old)
    np.savez_compressed(dst, **old)
    print("updated", os.path.normpath(dst), os.path.getsize(dst), "bytes")
    sys.exit(0)


def fasta(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">"))


seqs = {os.path.basename(p)[:-3]: fasta(p) for p in sorted(glob.glob(REF + "/data/*.fa"))}
rng = random.Random(20261004)
for n in (1, 2, 3, 4, 5, 6, 7, 8, 12, 17, 31, 32, 33, 40, 64, 65, 100):
    seqs["rnd%d" % n] = "".join(rng.choice("ACGU") for _ in range(n))
seqs["mixedTN"] = "acgTNNgcauGGGAAACCCuuuXgcgc"          # lower case, T and unknown letters -> unpairable
seqs["polyA"] = "A" * 24                                    # nothing can pair
seqs["gchelix"] = "GGGGGGGGGGAAAACCCCCCCCCC"                 # one dominant helix
s200 = random_pair(200)[0]
s500a, s500b = random_pair(500)
seqs["mt200"] = s200
TABLES_FOR = {"DIS", "Tar", "Tarstar", "rnd5", "rnd8", "rnd17", "rnd33", "mixedTN", "gchelix"}

out = {}
names = []
for nm, s in seqs.items():
    n = len(s)
    T = (n + 1) * (n + 2) // 2
    post = np.zeros(T)
    f5 = np.zeros(2 * (n + 1))
    tabs = np.zeros(6 * T) if nm in TABLES_FOR else None
    z = lib.ref_inference(s.encode(), 0, post.ctypes.data, tabs.ctypes.data if tabs is not None else None, f5.ctypes.data)
    zf = lib.ref_inference(s.encode(), 1, None, None, None)
    out["mc/%s/seq" % nm] = np.array(s)
    out["mc/%s/logZ" % nm] = np.array(z)
    out["mc/%s/logZ_float" % nm] = np.array(zf)
    out["mc/%s/post" % nm] = post
    out["mc/%s/f5" % nm] = f5
    if tabs is not None:
        out["mc/%s/tables" % nm] = tabs
    names.append(nm)
    print("mc %-10s n=%4d logZ=%.12f sum=%.10f" % (nm, n, z, post.sum()))

# n=500 (BASELINE.json config 3): scalars + a strided sample of the posterior
for nm, s in (("mt500a", s500a), ("mt500b", s500b)):
    n = len(s)
    T = (n + 1) * (n + 2) // 2
    post = np.zeros(T)
    z = lib.ref_inference(s.encode(), 0, post.ctypes.data, None, None)
    out["mc500/%s/seq" % nm] = np.array(s)
    out["mc500/%s/logZ" % nm] = np.array(z)
    out["mc500/%s/post_sum" % nm] = np.array(post.sum())
    idx = np.flatnonzero(post > 1e-4)
    out["mc500/%s/idx" % nm] = idx.astype(np.int64)
    out["mc500/%s/val" % nm] = post[idx]
    print("mc500 %s logZ=%.9f sum=%.9f nnz(>1e-4)=%d" % (nm, z, post.sum(), idx.size))

# n=1000 (several far blocks per tile row in the GPU kernels): scalars + sparse posterior
s1000 = random_pair(1000)[0]
post = np.zeros(1001 * 1002 // 2)
z = lib.ref_inference(s1000.encode(), 0, post.ctypes.data, None, None)
out["mc1000/seq"] = np.array(s1000)
out["mc1000/logZ"] = np.array(z)
out["mc1000/post_sum"] = np.array(post.sum())
idx = np.flatnonzero(post > 1e-4)
out["mc1000/idx"] = idx.astype(np.int64)
out["mc1000/val"] = post[idx]
print("mc1000 logZ=%.9f sum=%.9f nnz=%d" % (z, post.sum(), idx.size))

# n=2000 (BASELINE config 4): scalars + sparse posteriors of the mt19937(12345) pair; ~3 min of reference CPU time
if os.environ.get("GOLDEN_SKIP_2000") != "1":
    a2000, b2000 = random_pair(2000)
    post = np.zeros(2001 * 2002 // 2)
    z = lib.ref_inference(a2000.encode(), 0, post.ctypes.data, None, None)
    out["mc2000/logZ"] = np.array(z)
    out["mc2000/post_sum"] = np.array(post.sum())
    idx = np.flatnonzero(post > 1e-3)
    out["mc2000/idx"] = idx.astype(np.int64)
    out["mc2000/val"] = post[idx]
    print("mc2000 logZ=%.9f sum=%.9f nnz=%d" % (z, post.sum(), idx.size))
    hp2000 = np.zeros(2001 * 2001); z2 = np.zeros(2)
    lib.ref_duplex(a2000.encode(), b2000.encode(), 0, hp2000.ctypes.data, None, None, z2.ctypes.data)
    out["dx2000/logZ2"] = z2
    out["dx2000/post_sum"] = np.array(hp2000.sum())
    idx = np.flatnonzero(hp2000 > 1e-4)
    out["dx2000/idx"] = idx.astype(np.int64)
    out["dx2000/val"] = hp2000[idx]
    print("dx2000 logZ=%.9f sum=%.9f nnz=%d" % (z2[0], hp2000.sum(), idx.size))
    golden_2000b(out)

pairs = [("DIS", "DIS"), ("CopA", "CopT"), ("IncRNA54", "RepZ"), ("MicA", "ompA"), ("OxyS", "fhlA"),
         ("R1inv", "R2inv"), ("RyhB", "SodB"), ("Tar", "Tarstar"), ("rnd1", "rnd3"), ("rnd5", "rnd8"),
         ("rnd17", "rnd33"), ("rnd40", "rnd64"), ("rnd65", "rnd31"), ("mixedTN", "rnd17"), ("polyA", "polyA"),
         ("gchelix", "gchelix"), ("rnd100", "mt200")]
DT_FOR = {("DIS", "DIS"), ("Tar", "Tarstar"), ("rnd5", "rnd8"), ("rnd17", "rnd33"), ("mixedTN", "rnd17")}
pnames = []
for a, b in pairs:
    s1, s2 = seqs[a], seqs[b]
    S = (len(s1) + 1) * (len(s2) + 1)
    post = np.zeros(S)
    ins = np.zeros(S) if (a, b) in DT_FOR else None
    outs = np.zeros(S) if (a, b) in DT_FOR else None
    z2 = np.zeros(2)
    lib.ref_duplex(s1.encode(), s2.encode(), 0, post.ctypes.data,
                   ins.ctypes.data if ins is not None else None,
                   outs.ctypes.data if outs is not None else None, z2.ctypes.data)
    key = "%s+%s" % (a, b)
    out["dx/%s/logZ2" % key] = z2
    out["dx/%s/post" % key] = post
    if ins is not None:
        out["dx/%s/inside" % key] = ins
        out["dx/%s/outside" % key] = outs
    pnames.append(key)
    print("dx %-18s logZ=%.12f / %.12f  max=%.9f" % (key, z2[0], z2[1], post.max()))
# n=500 duplex: scalars + sparse sample
S = 501 * 501
post = np.zeros(S); z2 = np.zeros(2)
lib.ref_duplex(s500a.encode(), s500b.encode(), 0, post.ctypes.data, None, None, z2.ctypes.data)
out["dx500/logZ2"] = z2
out["dx500/post_sum"] = np.array(post.sum())
idx = np.flatnonzero(post > 1e-6)
out["dx500/idx"] = idx.astype(np.int64)
out["dx500/val"] = post[idx]
print("dx500 logZ=%.9f sum=%.9f nnz=%d" % (z2[0], post.sum(), idx.size))

# z-score shuffles: the reference's own ushuffle.c (oracle/_ref/libref_ushuffle.so) driven exactly as
# src/ractip.cpp:1636-1643 does (srandom(seed); shuffle(s1) then shuffle(s2) per iteration, k=2)
ush = ctypes.CDLL(os.path.join(here, "_ref", "libref_ushuffle.so"))
libc = ctypes.CDLL(None)
ush.shuffle.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
for tag, a, b, mode, seed, num in (("OxyS+fhlA/12/1", "OxyS", "fhlA", 12, 1, 12), ("DIS+Tar/1/7", "DIS", "Tar", 1, 7, 6),
                                   ("R1inv+R2inv/2/3", "R1inv", "R2inv", 2, 3, 6), ("rnd2+rnd3/12/5", "rnd2", "rnd3", 12, 5, 4)):
    s1, s2 = seqs[a], seqs[b]
    t1, t2 = ctypes.create_string_buffer(s1.encode()), ctypes.create_string_buffer(s2.encode())
    libc.srandom(seed)
    rows = []
    for it in range(num):
        if mode in (1, 12):
            ush.shuffle(s1.encode(), t1, len(s1), 2)
        if mode in (2, 12):
            ush.shuffle(s2.encode(), t2, len(s2), 2)
        rows.append(t1.value.decode() + "|" + t2.value.decode())
    out["zs/%s" % tag] = np.array(rows)
    print("zs", tag, rows[0][:40])

out["mc_names"] = np.array(names)
out["dx_names"] = np.array(pnames)
dst = os.path.join(here, "..", "tests", "golden", "contrafold_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", os.path.normpath(dst), os.path.getsize(dst), "bytes")
