#!/usr/bin/env python3
"""Developer aid: compile fpt_kernels.hip to gfx950 assembly and print, per kernel, registers,
scratch, LDS and static instruction counts (VALU / SALU / LDS / VMEM).  Usage:
    python tools/kernel_stats.py [substring-of-kernel-name [source-file]]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/fpt_kernels.s"
SRC = sys.argv[2] if len(sys.argv) > 2 else "fpt_kernels.hip"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm",
                       "-disable-machine-licm", "-S", "--cuda-device-only", "-x", "hip",
                       os.path.join(ROOT, "footprint_tools_amd", "csrc", SRC), "-o", out],
                      stderr=subprocess.DEVNULL)
txt = open(out).read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
meta = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    d = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
    meta[m.group(1)] = d
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
    name = m.group(1)
    if flt not in name or name not in meta:
        continue
    c = collections.Counter()
    for l in m.group(2).split("\n"):
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        op = s.split()[0]
        c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
          else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += 1
    d = meta[name]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print("%-70s vgpr %3s sgpr %3s scratch %4s lds %6s | static valu %5d salu %5d lds %4d vmem %3d" % (
        dem[:70], d.get("vgpr_count"), d.get("sgpr_count"), d.get("private_segment_fixed_size"),
        d.get("group_segment_fixed_size"), c["valu"], c["salu"], c["lds"], c["vmem"]))
