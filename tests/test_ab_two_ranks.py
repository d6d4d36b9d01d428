"""Two ranks on ONE GPU through the librccl stand-in (tests/fakerccl): everything of the sharded job
but RCCL itself with a rank > 0 -- TrackComm's rendezvous, shard offsets, the all-gather and the gather
to a root in place and out of place, their asynchronous forms, `bench.py --gpus 2`.  Sorts early on
purpose (after test_aa_): the children are started before this process has touched the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

from .conftest import ROOT

FAKE = os.path.join(ROOT, "tests", "fakerccl", "libfakerccl.so")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake():
    """the stand-in library, built on the spot if the tree came without it (hipcc is in the image; this process
    has not touched the GPU)"""
    src = os.path.join(ROOT, "tests", "fakerccl", "fakerccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", FAKE])
    return FAKE


def _spawn(argv, world, extra_env, timeout=900):
    _fake()
    port = str(_port())
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   FPT_RCCL_LIB=FAKE, HSA_ENABLE_IPC_MODE_LEGACY="0", FPT_COMM_TIMEOUT_S="120", **extra_env)
        procs.append(subprocess.Popen([sys.executable] + argv, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o.decode(), e.decode()))
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["ragged", "uniform"])
def test_two_ranks_one_gpu_collectives(shape):
    outs = _spawn([os.path.join(ROOT, "tests", "two_rank_worker.py")], 2, {"FPT_TWO_RANK_SHAPE": shape})
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and ("RANK %d OK" % r) in o, "rank %d: rc %d\n%s\n%s" % (r, rc, o[-1500:], e[-3000:])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8])
def test_many_ranks_one_gpu_collectives(world):
    """the same at N = 4 and 8 with ragged shards: shard-offset arithmetic with eight slices, the in-place
    offsets, gathers to every root in turn, `wait(back=1)` with seven peers, and what the communicator
    reports about the job (ncclCommCount / ncclCommUserRank per rank)"""
    bad = None
    for attempt in range(2):
        outs = _spawn([os.path.join(ROOT, "tests", "two_rank_worker.py")], world,
                      {"FPT_TWO_RANK_SHAPE": "ragged", "FPT_TWO_RANK_INTERVALS": "1601"})
        bad = ["rank %d: rc %d\n%s\n%s" % (r, rc, o[-300:], e[-700:]) for r, (rc, o, e) in enumerate(outs)
               if rc != 0 or ("RANK %d OK" % r) not in o]
        if not bad:
            break
        # One more try, and only for the stand-in's own rendezvous: eight processes opening each other's HIP IPC
        # handles on one GPU have been seen to miss its 60 s barrier once (a test-infrastructure timeout, every
        # rank's tail is printed); anything else -- a wrong track, a failed assertion -- fails at once
        print("attempt %d failed:\n%s" % (attempt, "\n".join(bad)))
        if not any("fakerccl" in b and "timed out" in b for b in bad):
            break
    assert not bad, "\n".join(bad)  # (every failing rank: the first to give up is rarely the one that went wrong)


@pytest.mark.gpu
@pytest.mark.parametrize("assembly", ["allgather", "gather"])
def test_bench_two_ranks_one_gpu(assembly):
    """bench.py --gpus 2 end to end (config 4: ONE global ragged list cut in two, the track assembled):
    rank 0's line carries the multi-GPU block, and its spot check reads the OTHER rank's slice of the
    assembled track"""
    outs = _spawn([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--intervals", "20000", "--steps", "3",
                   "--warmup", "1", "--share-gpu", "--assembly", assembly], 2, {})
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, "rank %d: rc %d\n%s\n%s" % (r, rc, o[-1500:], e[-3000:])
    lines = [ln for ln in outs[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][1].splitlines() if ln.startswith("{")]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity"]["exp_bit_exact"] is True and d["parity"]["p_max_rel_err"] < 1e-6
    assert d["parity"]["gathered_last_rank_p_max_rel_err"] < 1e-6
    mg = d["multi_gpu"]
    assert mg["assembly"] == assembly and mg["steps"] == 3
    assert mg["scan_only"]["value"] > 0 and mg["with_assembly"]["value"] > 0 and mg["with_assembly"]["overlapped"] is True
    # the communicator's own account of the job: two ranks, each reporting itself
    assert mg["rccl_ranks"] == 2 and [r_["rccl_user_rank"] for r_ in mg["ranks"]] == [0, 1]
    assert all(r_["pci_bus_id"] for r_ in mg["ranks"])


@pytest.mark.gpu
def test_bench_launches_two_ranks_itself():
    """`python3 bench.py --gpus 2 ...` as ONE command with no launcher around it (what the driver's scaling run is
    most likely to issue): the parent starts both ranks before touching the GPU (bench.launch_ranks), relays rank 0's
    one line and exits 0; the line is a two-rank job by the communicator's own account"""
    _fake()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "FPT_COMM_FILE")}
    env.update(FPT_RCCL_LIB=FAKE, HSA_ENABLE_IPC_MODE_LEGACY="0", FPT_COMM_TIMEOUT_S="120", FPT_LAUNCH_TIMEOUT_S="600")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--intervals", "20000",
                          "--steps", "3", "--warmup", "1", "--share-gpu"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["parity"]["exp_bit_exact"] is True and d["parity"]["p_max_rel_err"] < 1e-6
    mg = d["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and [r_["rccl_user_rank"] for r_ in mg["ranks"]] == [0, 1]
    assert 0 < mg["expected_value_vs_linear"] < 1 and 0 < mg["value_over_scan_only"] <= 1
