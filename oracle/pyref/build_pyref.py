#!/usr/bin/env python3
"""DEV-CONTAINER TOOLING (test infrastructure): build the genuine reference's
Cython extensions so that the real `footprint_tools` can be imported HERE to
generate golden vectors (tests/golden/make_golden.py).

Nothing is copied: every .pyx / .c is compiled where it lies under
/root/reference with explicit `cython` + `gcc` calls (the flags distutils would
use: -O2 -fwrapv, no -march, no fast-math) and all outputs go to /tmp/fpt_pyref.
The result cannot travel to the GPU box; only the vectors it produces do.
"""
import glob
import os
import subprocess
import sys
import sysconfig

import numpy as np

REF = os.environ.get("FPT_REFERENCE", "/root/reference")
OUT = os.environ.get("FPT_PYREF_OUT", "/tmp/fpt_pyref")

MODULES = [
    "footprint_tools/modeling/predict.pyx",
    "footprint_tools/modeling/dispersion.pyx",
    "footprint_tools/stats/utils.pyx",
    "footprint_tools/stats/windowing.pyx",
    "footprint_tools/stats/distributions/nbinom.pyx",
]


def run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present")
    os.makedirs(os.path.join(OUT, "obj"), exist_ok=True)
    cflags = ["-O2", "-fwrapv", "-fPIC", "-w", "-fno-strict-aliasing"]
    # vendored hcephes -> static objects (setup.py:20-22 builds the same list as a clib)
    objs = []
    for src in sorted(glob.glob(os.path.join(REF, "hcephes/src/**/*.c"), recursive=True)):
        o = os.path.join(OUT, "obj", src[len(REF) + 1:].replace("/", "_") + ".o")
        if not os.path.exists(o):
            run(["gcc", *cflags, "-I", os.path.join(REF, "hcephes/include"), "-c", src, "-o", o])
        objs.append(o)
    lib = os.path.join(OUT, "libhcephes.a")
    if not os.path.exists(lib):
        run(["ar", "rcs", lib, *objs])

    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    pyinc = sysconfig.get_paths()["include"]
    for rel in MODULES:
        stem = rel[:-4]
        cfile = os.path.join(OUT, "c", stem + ".c")
        so = os.path.join(OUT, "pkg", stem + suffix)
        if os.path.exists(so):
            continue
        os.makedirs(os.path.dirname(cfile), exist_ok=True)
        os.makedirs(os.path.dirname(so), exist_ok=True)
        run([sys.executable, "-m", "cython", "-3", "-I", REF, os.path.join(REF, rel), "-o", cfile])
        run(["gcc", *cflags, "-shared", "-DNPY_NO_DEPRECATED_API=0", "-I", pyinc,
             "-I", np.get_include(), "-I", os.path.join(REF, "hcephes/include"),
             "-I", os.path.join(REF, os.path.dirname(rel)), cfile, lib, "-lm", "-o", so])
    print("built into", OUT)


if __name__ == "__main__":
    main()
