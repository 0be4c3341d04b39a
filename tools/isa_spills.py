#!/usr/bin/env python3
"""List the scratch (spill) instructions of a kernel with the loop they sit in.
usage: tools/isa_spills.py <file.hip> <mangled-kernel-substring> [extra hipcc flags...]"""
import re, subprocess, sys, tempfile, os
src, kern = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-fno-fast-math",
                "-S", "--cuda-device-only", "-o", out, src] + extra, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
os.unlink(out)
inside = False
label, ctx = "", ""
nv = {}
for i, l in enumerate(lines):
    if re.match(r"^_Z\w*:", l):
        inside = kern in l
        label, ctx = "entry", ""
        continue
    if not inside:
        continue
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
    if m:
        label, ctx = m.group(1), (m.group(2) or "").strip()
        # following comment lines carry loop info
        j = i + 1
        while j < len(lines) and lines[j].strip().startswith(";"):
            ctx += " " + lines[j].strip()
            j += 1
    if "scratch_" in l:
        print(f"{label:12s} {ctx[:70]:70s} {l.strip()[:60]}")
    m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", l)
    if m: print("next_free_vgpr", m.group(1))
    m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", l)
    if m: print("scratch bytes", m.group(1)); inside = False
