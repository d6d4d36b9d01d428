"""The parity suite once more under the library's diagnostic kernel selections, each as a child process (started
before this process has touched the GPU: the file sorts ahead of the suites that do): FPT_SCAN_LEAN=0 -- the general
kernel's memo-only instance as the first pass instead of k_scan_lean; FPT_TABLE_LDS=1 -- the 6-mer table staged in
LDS per workgroup (BASELINE.json's wording; the default gathers it through L1 / L2, measured faster);
FPT_LEAN_TAB=0 -- phase E's normal cdf from the Horner chain instead of the LDS table of round 6.  Every parity
test must hold in every mode: the modes are what the A/B measurements of DESIGN.md compare."""
import os
import subprocess
import sys

import pytest

from .conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["FPT_SCAN_LEAN=0", "FPT_TABLE_LDS=1", "FPT_LEAN_TAB=0"])
def test_parity_suite_under_kernel_mode(mode):
    key, val = mode.split("=")
    env = dict(os.environ, **{key: val})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FPT_FUZZ_SEEDS"] = "4"  # (the fuzz families at four seeds each: the default ten run in the main suite)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                          "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = out.stdout.decode()[-3000:]
    assert out.returncode == 0, "%s:\n%s" % (mode, tail)
    assert " passed" in tail and "failed" not in tail, tail
