#!/usr/bin/env python3
"""tools/isa_serial.py FILE.hip [...]: compile for gfx950 and list, per kernel, the places where a global load is waited for with
vmcnt(0) within a few instructions of its issue (a load the compiler sank into a branch: one exposed memory round trip each) and the
scratch (spill) traffic.  A static screen for the pattern that cost the duplex and outside strip kernels a third of their time."""
import re, subprocess, sys, tempfile, os
for src in sys.argv[1:]:
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    name, lines = None, []
    kernels = {}
    for ln in open(out):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1); kernels[name] = []
        elif name:
            kernels[name].append(ln)
            if "s_endpgm" in ln: name = None
    os.unlink(out)
    for k, body in kernels.items():
        serial, spills, last_load = 0, 0, -100
        for idx, ln in enumerate(body):
            if re.search(r"\b(global|flat|buffer)_load", ln): last_load = idx
            if "scratch_" in ln: spills += 1
            if re.search(r"s_waitcnt vmcnt\(0\)", ln) and idx - last_load <= 6: serial += 1
        if serial or spills:
            d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%-28s %3d serialized load(s) %3d scratch instr  %s" % (os.path.basename(src), serial, spills, d[:110]))
