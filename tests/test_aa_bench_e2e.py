"""bench.py end to end as child processes, on the GPU box: the sharded job's path -- shard the
global interval list, scan, ONE RCCL all-gather of the track (here on a one-rank communicator, with
the ragged grouped-broadcast form forced) -- for BASELINE configs 4 and 5, and the posterior-caller
leg of config 5.  The file sorts first on purpose: the children are started before this process
has touched the GPU (a process that has initialised the GPU must not fork/exec on the GPU pool)."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import ROOT


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]   # ONE JSON line on stdout (the RCCL banner goes to stderr)
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("ragged_form", ["0", "1"])
def test_bench_config4_allgather_one_rank(ragged_form):
    d = _bench(["--gpus", "1", "--config", "4", "--intervals", "30000", "--steps", "3", "--warmup", "1", "--allgather",
                "--no-cpu-baseline", "--no-other-mode"], {"FPT_COMM_RAGGED": ragged_form})
    assert d["n_gpus"] == 1 and d["config"]["allgather_p_track"] is True
    mg = d["multi_gpu"]
    assert mg["allgather_bytes_per_rank"] == 8 * sum(mg["bases_per_rank"]) and mg["allgather_s"] > 0
    # the spot check reads the GATHERED track: it holds this rank's p-values, equal to the oracle's
    p = d["parity"]
    assert p["exp_bit_exact"] is True and p["p_max_rel_err"] < 1e-6 and p["winp_max_rel_err"] < 1e-6
    assert d["roofline"]["kernel_ms"] > 0 and d["value"] > 1e9


@pytest.mark.gpu
def test_bench_config5_allgather_and_posterior_one_rank():
    d = _bench(["--gpus", "1", "--config", "5", "--intervals", "8000", "--steps", "2", "--warmup", "1", "--allgather",
                "--no-cpu-baseline", "--no-other-mode"], {"FPT_COMM_RAGGED": "1"})
    p = d["parity"]
    assert p["exp_bit_exact"] is True and p["p_max_rel_err"] < 1e-6
    assert p["efdr_max_abs_err"] <= 2.5 / (50 * 100)   # the gathered FDR track against the oracle's sampler
    assert d["config"]["empirical_fdr_null_draws_per_base"] == 100
    post = d["posterior"]
    assert post["n_datasets"] == 8 and post["parity_ok"] is True and post["zero_division_flags"] == 0
    assert post["value"] > 1e7


@pytest.mark.gpu
def test_bench_default_line_carries_every_configuration():
    """the default invocation's line (here on small interval counts): config 3 at the top level and a `configs`
    object with configs 2, 3, 4 and 5, each with its own step time, parity check and a roofline block for its own
    dominant kernel -- config 5's names the FDR draw kernels and an issue bound, not the scan's HBM bytes"""
    d = _bench(["--intervals", "20000", "--leg-intervals", "20000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                "--no-other-mode", "--no-heavy", "--no-traffic-probe"], {})
    assert d["config"]["workload"].startswith("20000x1kb_5scales") and set(d["configs"]) == {"2", "3", "4", "5"}
    for cid, leg in d["configs"].items():
        assert "error" not in leg, (cid, leg)
        assert leg["ms_per_step"] > 0 and leg["value"] > 1e8 and leg["parity"]["exp_bit_exact"] is True, cid
        assert leg["parity"]["p_max_rel_err"] < 1e-6 and leg["roofline"]["bound"] in ("hbm", "valu+lds issue"), cid
    assert d["configs"]["3"]["ms_per_step"] == d["ms_per_step"]
    c5 = d["configs"]["5"]
    assert c5["roofline"]["bound"] == "valu+lds issue" and "k_fdr" in c5["roofline"]["kernel"]
    assert c5["roofline"]["scan"]["bound"] == "hbm" and c5["fdr"]["ms_per_call"] > 0 and c5["parity"]["efdr_max_abs_err"] <= 2.5 / (50 * 100)
    assert c5["posterior"]["parity_ok"] is True
    # (with --leg-intervals the counters are not collected: the fractions are None there, the keys are in the line)
    assert "ta_busy" in c5["roofline"] and "lds_busy" in c5["roofline"]
    assert d["configs"]["4"]["workload"].startswith("20000xragged")
    # the host-arrays leg (numpy in / numpy out through FootprintScanner.scan, config 2's size, PCIe included)
    h = d["host_arrays"]
    assert "error" not in h, h
    assert h["parity"]["exp_bit_exact"] is True and h["parity"]["p_max_rel_err"] < 1e-6 and h["parity"]["pinned_equals_pageable_bitwise"] is True
    assert h["pinned"]["arrays_page_locked"] is True and h["pageable"]["arrays_page_locked"] is False
    assert h["pinned"]["value"] > 1e8 and h["pageable"]["value"] > 1e8
    # the CPU baseline's figures as plain scalars are absent here (--no-cpu-baseline); the multi-GPU block is None at N = 1
    assert d["multi_gpu"] is None
    # the compact summary that closes the line repeats every configuration's figures
    sm = d["summary"]
    assert list(d)[-1] == "summary" and set(sm["configs"]) == {"2", "3", "4", "5"} and sm["degraded"] == 0
    assert all(c["exp_bit_exact"] is True and c["ms_per_step"] > 0 for c in sm["configs"].values())
    assert sm["host_arrays_bases_per_s"]["parity_ok"] is True
    # no child run of this invocation failed silently
    assert d["degraded"] == [] and not any("degraded" in leg for leg in d["configs"].values())
