"""world_size-2 check of the multi-GPU path's host logic on CPU (gloo): interval sharding by
padded bases + the single all-gather that re-assembles the per-base p-value track.  The
per-shard compute is stood in for by the CPU oracle (test infrastructure) -- the product's
scan needs a GPU -- so this covers exactly what differs between N=1 and N>1."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from .conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["FPT_ROOT"])
from oracle import oracle
from footprint_tools_amd.scan import shard_intervals
from footprint_tools_amd.distributed import shard_offsets, shard_track_sizes
sys.path.insert(0, os.path.join(os.environ["FPT_ROOT"], "tests"))
from torch_gather import allgather_track, gather_track   # torch stand-ins of the two collectives (test infrastructure)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = np.load(os.path.join(os.environ["FPT_ROOT"], "tests/golden/kmer_probs.npz"))
lat = np.load(os.path.join(os.environ["FPT_ROOT"], "tests/golden/nb_lattice.npz"))
hw, shw, clip, scales = 5, 50, 0.01, np.array([3], np.int32)
pad = hw + shw
mode = os.environ["FPT_MODE"]
if mode == "uniform":
    lens = np.full(37, 200)
else:
    lens = np.clip(np.random.RandomState(3).lognormal(5, .6, 41).astype(int), 50, 900)
bounds = shard_intervals(lens if mode != "uniform" else (37, 200), world, pad)
sizes = shard_track_sizes(lens, bounds)

def run(first, last):
    ps = []
    pos = int((lens[:first] + 2 * pad + 7).sum())
    for L in lens[first:last]:
        l = int(L) + 2 * pad + 1
        cp, cm = oracle.synth_counts(4, pos, l, 0), oracle.synth_counts(4, pos, l, 1)
        sq = oracle.synth_bases(4, pos, l + 6)
        pos += l + 6
        ps.append(oracle.detect_batch(cp, cm, sq, 1, int(L), hw, shw, clip, g["table"], lat["mu_A"],
                                      lat["r_A"], scales)[2])
    return np.concatenate(ps) if ps else np.zeros(0)

a, b = bounds[rank]
local = torch.from_numpy(run(a, b))
full = allgather_track(local, sizes).numpy()
want = run(0, len(lens))
assert full.shape == want.shape and np.array_equal(full, want, equal_nan=True), "gathered track differs"
# the gather to the rank that writes (fpt_gather_track's shape): the whole track on the root only, every
# slice at shard_offsets(sizes)
for root in range(world):
    got = gather_track(local, sizes, root=root)
    if rank == root:
        assert np.array_equal(got.numpy(), want, equal_nan=True), "track gathered to rank %d differs" % root
        off = shard_offsets(sizes)
        assert int(off[-1]) == want.size and np.array_equal(got.numpy()[off[rank]:off[rank + 1]], local.numpy(), equal_nan=True)
    else:
        assert got is None

# the sharded detect driver's host logic (range per rank, row gather, records of ALL intervals in
# list order on every rank) with a CPU stand-in for the per-rank GPU driver
from footprint_tools_amd.distributed import sharded_deviation_stats

class Iv(object):
    def __init__(self, i, L):
        self.chrom, self.start, self.end, self.i = "c", 0, int(L), i

class StubStats(object):
    def __init__(self, intervals, read_func, fasta_func, bm, dm, batch_size=4096, **kw):
        self.intervals, self.padding = list(intervals), pad
    def compute(self, indices):
        out = []
        for i in indices:
            p = run(i, i + 1)
            out.append({"interval": self.intervals[i], "stats": np.column_stack([p, p * 2, p + 1, p, p])})
        return out

def gather_rows(local, counts):
    k = local.shape[1]
    flat = allgather_track(torch.from_numpy(np.ascontiguousarray(local).ravel()), [c * k for c in counts])
    return flat.numpy().reshape(-1, k)

ivs = [Iv(i, L) for i, L in enumerate(lens)]
recs = sharded_deviation_stats(ivs, None, None, None, True, gather=gather_rows, rank=rank, world=world,
                               batch_size=7, stats_cls=StubStats)
assert len(recs) == len(ivs)
for i, rec in enumerate(recs):
    p = run(i, i + 1)
    assert rec["interval"] is ivs[i] and rec["stats"].shape == (int(lens[i]), 5)
    assert np.array_equal(rec["stats"][:, 0], p, equal_nan=True) and np.array_equal(rec["stats"][:, 2], p + 1, equal_nan=True)
if rank == 0:
    print("OK", mode, bounds, sizes)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode", ["uniform", "ragged"])
def test_shard_and_allgather_two_ranks_gloo(tmp_path, mode):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FPT_ROOT=ROOT, FPT_MODE=mode, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK " + mode in out.stdout


@pytest.mark.parametrize("world", [4, 8])
def test_shard_and_allgather_many_ranks_gloo(tmp_path, world):
    """the same with 4 and 8 ranks (the node's N = 4 and N = 8): ragged shards, the all-gather, a gather to every
    root in turn, and the sharded driver's records -- 41 intervals over 8 ranks leaves shards of 4 - 6 intervals"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FPT_ROOT=ROOT, FPT_MODE="ragged", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK ragged" in out.stdout


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_intervals_properties(world):
    """`scan.shard_intervals` on the whole-genome shape (3.5 M lognormal lengths) and on degenerate lists: the
    ranges are contiguous, in order and cover the list; every rank's padded bases are within one interval of the
    ideal share; fewer intervals than ranks leaves the surplus ranks empty; offsets and sizes agree with them"""
    from footprint_tools_amd.distributed import shard_offsets, shard_track_sizes
    from footprint_tools_amd.scan import shard_intervals
    pad = 55
    lens = np.clip(np.random.RandomState(4).lognormal(4.9, 0.62, 3500000), 50, 2000).astype(np.int64)
    b = shard_intervals(lens, world, pad)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == lens.size
    assert all(b[r][1] == b[r + 1][0] for r in range(world - 1)) and all(x <= y for x, y in b)
    cost = lens + 2 * pad + 1
    share = cost.sum() / world
    got = np.array([cost[x:y].sum() for x, y in b])
    assert np.all(np.abs(got - share) <= 2 * cost.max()), (got - share)
    sizes = shard_track_sizes(lens, b)
    off = shard_offsets(sizes)
    assert sum(sizes) == lens.sum() and off[0] == 0 and off[-1] == lens.sum() and np.all(np.diff(off) == sizes)
    # a uniform batch given as (n, L), and lists shorter than the number of ranks
    bu = shard_intervals((1000003, 500), world, pad)
    assert bu[-1][1] == 1000003 and max(y - x for x, y in bu) - min(y - x for x, y in bu) <= 1
    for n in (0, 1, world - 1):
        bs = shard_intervals(np.full(max(n, 0), 100), world, pad)
        assert len(bs) == world and bs[-1][1] == n and sum(y - x for x, y in bs) == n


def test_id_rendezvous_ignores_leftovers(tmp_path):
    """The file rendezvous of TrackComm (rank 0 offers the 128-byte communicator id, the others wait
    for it): a leftover of an aborted run under the same name is never taken for the id -- an
    offer only counts while rank 0 keeps touching it -- wrong sizes are ignored, a withdrawn offer
    is gone, and successive communicators of one process use different names."""
    import threading
    import time
    from footprint_tools_amd import distributed as D
    path = str(tmp_path / "fpt_comm.id")
    old, new = bytes(range(128)), bytes(reversed(range(128)))
    with open(path, "wb") as f:           # what an aborted run left behind
        f.write(old)
    os.utime(path, (time.time() - 60, time.time() - 60))
    assert D._read_fresh_id(path) is None
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("id", D._await_id(path, 20.0)))
    t.start()
    time.sleep(0.3)
    assert "id" not in got                 # still waiting: the stale file does not count
    offer = D._id_offer(path, new)
    t.join(10)
    assert got.get("id") == new
    time.sleep(3 * D._BEAT_S)              # the offer stays fresh while it stands
    assert D._read_fresh_id(path) == new
    offer.withdraw()
    assert not os.path.exists(path) and D._read_fresh_id(path) is None
    with open(path, "wb") as f:            # a damaged file
        f.write(new[:100])
    assert D._read_fresh_id(path) is None
    with pytest.raises(TimeoutError):
        D._await_id(path, 0.2)
    a = D._id_path()
    D._comm_counter[0] += 1
    try:
        assert D._id_path() != a
    finally:
        D._comm_counter[0] -= 1


LAUNCHED = r'''
import os, sys, time
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["FPT_COMM_FILE"]
open(os.path.join(sys.argv[1], "rank%d" % r), "w").write(" ".join(
    [str(os.getpid()), os.environ["WORLD_SIZE"], os.environ["MASTER_PORT"], os.environ["FPT_COMM_FILE"]] + sys.argv[2:]))
mode = sys.argv[2]
if mode == "fail" and r == 3:
    time.sleep(0.5)
    sys.exit(3)
if mode in ("fail", "hang"):
    time.sleep(600)
print('{"rank": %d}' % r)
'''


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fpt_bench_for_tests", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:  # a zombie of another parent counts as gone
        return open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_bench_launches_its_own_ranks(tmp_path, capfd):
    """`python3 bench.py --gpus 8` with no launcher around it starts the eight ranks itself (bench.launch_ranks;
    the reference's counterpart is batch_iter(num_workers=n), cli/detect.py:394): every rank gets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT and one rendezvous file name, the arguments arrive unchanged, only rank
    0's standard output reaches the caller's.  A rank that exits 3 stops the job within seconds with a non-zero code
    and no process left behind; so does a job that outlives its limit."""
    import time
    bench = _load_bench()
    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED)
    d = tmp_path / "ok"
    d.mkdir()
    rc = bench.launch_ranks(8, [str(d), "ok", "--gpus", "8", "--steps", "2"], program=str(script))
    out = capfd.readouterr().out
    assert rc == 0 and out.strip() == '{"rank": 0}'
    seen = [open(d / ("rank%d" % r)).read().split() for r in range(8)]
    assert all(s[1] == "8" and s[5:] == ["--gpus", "8", "--steps", "2"] for s in seen)
    assert len(set(s[2] for s in seen)) == 1 and len(set(s[3] for s in seen)) == 1
    assert not os.path.exists(seen[0][3])

    d = tmp_path / "fail"
    d.mkdir()
    t0 = time.time()
    rc = bench.launch_ranks(8, [str(d), "fail"], program=str(script))
    assert rc == 3 and time.time() - t0 < 20
    pids = [int(open(d / ("rank%d" % r)).read().split()[0]) for r in range(8)]
    time.sleep(0.2)
    assert not [p for p in pids if _alive(p)]

    d = tmp_path / "hang"
    d.mkdir()
    t0 = time.time()
    rc = bench.launch_ranks(2, [str(d), "hang"], program=str(script), timeout_s=2.0)
    assert rc == 124 and time.time() - t0 < 20
    pids = [int(open(d / ("rank%d" % r)).read().split()[0]) for r in range(2)]
    time.sleep(0.2)
    assert not [p for p in pids if _alive(p)]


def test_bench_gpus_flag_reaches_the_launcher(tmp_path):
    """the command the driver would run: `python3 bench.py --gpus 8 ...` as ONE process.  Here (no GPU) the ranks
    cannot make a context, so each fails -- what is checked is that the parent launches rather than refuses,
    reports the failing rank and exits non-zero promptly."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["FPT_LAUNCH_TIMEOUT_S"] = "120"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "4", "--intervals", "1000",
                          "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the launched ranks run (covered by the -m gpu test)")
    assert out.returncode != 0 and "must be launched" not in out.stderr
    assert "exited with code" in out.stderr
