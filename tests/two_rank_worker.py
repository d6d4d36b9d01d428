"""One rank of a job whose ranks SHARE GPU 0 (started by tests/test_ab_two_ranks.py and by hand:
RANK / WORLD_SIZE / MASTER_PORT in the environment, FPT_RCCL_LIB = tests/fakerccl/libfakerccl.so).
RCCL refuses two ranks on one device; with the stand-in bound in its place everything of the sharded
job except RCCL itself runs with a rank > 0: the rendezvous of TrackComm, the shard offsets, both
collectives in place and out of place, on the compute stream and on the communicator's own stream
with two track buffers in turn, the row gather, barrier and max.  Every rank checks the assembled
track against its OWN unsharded scan of the whole batch, bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from footprint_tools_amd import _lib  # noqa: E402
from footprint_tools_amd.distributed import TrackComm, shard_offsets, shard_track_sizes  # noqa: E402
from footprint_tools_amd.scan import DeviceArray, FootprintScanner, shard_intervals  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ragged = os.environ.get("FPT_TWO_RANK_SHAPE", "ragged") == "ragged"
    g = np.load(os.path.join(ROOT, "tests", "golden", "kmer_probs.npz"))
    lat = np.load(os.path.join(ROOT, "tests", "golden", "nb_lattice.npz"))
    DM = type("DM", (), dict(mu_params=lat["mu_A"], r_params=lat["r_A"]))
    hw, shw, clip, scales = 5, 50, 0.01, (3,)
    pad = hw + shw
    ctx = _lib.Context(0)  # every rank on GPU 0
    sc = FootprintScanner(g["table"], DM, hw, shw, clip, scales, ctx=ctx, nb_mode="memo")
    comm = TrackComm(ctx, rank, world, timeout_s=120.0)

    n_iv = int(os.environ.get("FPT_TWO_RANK_INTERVALS", "3001"))
    if ragged:
        lens = np.clip(np.random.RandomState(11).lognormal(4.9, 0.62, n_iv), 50, 2000).astype(np.int64)
    else:
        lens = np.full(n_iv - 1, 200, dtype=np.int64)   # (an even count: equal shards -> ncclAllGather)
        n_iv = lens.size
    bounds = shard_intervals(lens, world, pad)
    sizes = shard_track_sizes(lens, bounds)
    offs = shard_offsets(sizes)
    total_all = int(offs[-1])

    def scan_range(first, last, p_ptr):
        """the scan of intervals [first, last) of the global job; its p-values at p_ptr"""
        ln = lens[first:last]
        off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        tot, n = int(off[-1]), int(ln.size)
        n_counts, n_seq = sc.input_sizes(n, tot)
        pos_c = int((lens[:first] + 2 * pad + 1).sum())
        pos_s = int((lens[:first] + 2 * pad + 7).sum())
        d_cp, d_cm, d_sq = DeviceArray(ctx, n_counts * 8), DeviceArray(ctx, n_counts * 8), DeviceArray(ctx, n_seq)
        _lib.check(ctx.L.fpt_synth_dev(ctx.h, 3, pos_c, n_counts, d_cp.ptr, d_cm.ptr, pos_s, n_seq, d_sq.ptr))
        d_o = DeviceArray(ctx, 3 * tot * 8)
        d_off = DeviceArray(ctx, off.nbytes).upload(off)
        sc.scan_dev(n, d_cp.ptr, d_cm.ptr, d_sq.ptr, exp_out=d_o.ptr, obs_out=d_o.ptr + tot * 8, pval_out=p_ptr,
                    winp_out=d_o.ptr + 2 * tot * 8, interval_off_dev=d_off.ptr, interval_off_host=off)
        ctx.synchronize()
        for d in (d_cp, d_cm, d_sq, d_o, d_off):
            d.free()

    # the whole job on this rank alone: what the assembled track must equal
    d_full = DeviceArray(ctx, total_all * 8)
    scan_range(0, n_iv, d_full.ptr)
    want = d_full.download(np.float64, total_all)
    a, b = bounds[rank]
    my_off, my_n = int(offs[rank]), int(sizes[rank])

    def check(dev, what):
        got = dev.download(np.float64, total_all)
        assert np.array_equal(got, want, equal_nan=True), "rank %d: %s differs from the unsharded scan" % (rank, what)

    # 1. all-gather in place: the shard is scanned straight into its slice of the assembled track
    d_g = DeviceArray(ctx, total_all * 8).zero()
    scan_range(a, b, d_g.ptr + my_off * 8)
    comm.allgather_dev(d_g.ptr + my_off * 8, sizes, d_g.ptr)
    ctx.synchronize()
    check(d_g, "all-gather in place")
    # 2. out of place: the shard in a buffer of its own
    d_mine = DeviceArray(ctx, max(my_n, 1) * 8)
    scan_range(a, b, d_mine.ptr)
    d_g.zero()
    comm.allgather_dev(d_mine.ptr, sizes, d_g.ptr)
    ctx.synchronize()
    check(d_g, "all-gather out of place")
    # 3. gather to each rank in turn: the whole track on the root, nothing touched elsewhere
    for root in range(world):
        d_g.zero()
        ctx.synchronize()
        comm.gather_dev(d_mine.ptr, sizes, d_g.ptr if rank == root else None, root=root)
        ctx.synchronize()
        if rank == root:
            check(d_g, "gather to rank %d" % root)
        else:
            assert not d_g.download(np.float64, total_all).any(), "rank %d: a non-root buffer was written" % rank
        comm.barrier()
    # ... and in place on the root
    d_g.zero()
    scan_range(a, b, d_g.ptr + my_off * 8)
    comm.gather_dev(d_g.ptr + my_off * 8, sizes, d_g.ptr if rank == 0 else None, root=0)
    ctx.synchronize()
    if rank == 0:
        check(d_g, "gather in place")
    # 4. the asynchronous forms, two track buffers in turn: batch k's track travels on the communicator's
    #    stream while batch k + 1 is scanned; wait(back=1) before a buffer is scanned into again
    bufs = [DeviceArray(ctx, total_all * 8).zero(), DeviceArray(ctx, total_all * 8).zero()]
    for k in range(5):
        buf = bufs[k % 2]
        comm.wait(back=1)
        scan_range(a, b, buf.ptr + my_off * 8)  # (synchronises the host with the compute stream, not with the communicator's)
        if k % 2:
            comm.gather_dev_async(buf.ptr + my_off * 8, sizes, buf.ptr if rank == 0 else None, root=0)
        else:
            comm.allgather_dev_async(buf.ptr + my_off * 8, sizes, buf.ptr)
    comm.synchronize()
    check(bufs[0], "asynchronous all-gather")
    if rank == 0:
        check(bufs[1], "asynchronous gather")
    # 5. the host conveniences
    rows = np.column_stack([want[my_off:my_off + my_n], 2.0 * want[my_off:my_off + my_n]])
    full = comm.allgather_rows(rows, sizes)
    assert np.array_equal(full[:, 0], want, equal_nan=True) and np.array_equal(full[:, 1], 2.0 * want, equal_nan=True)
    assert comm.max_over_ranks(float(rank)) == float(world - 1)
    # 6. what the communicator itself reports about the job (ncclCommCount / ncclCommUserRank), gathered
    me = comm.info()
    assert me["world_size"] == world and me["rank"] == rank, me
    assert me["rccl_count"] == world and me["rccl_user_rank"] == rank, me
    assert len(me["pci_bus_id"].split(":")) == 3, me
    job = comm.job_info()
    assert [j["rccl_user_rank"] for j in job] == list(range(world)) and all(j["rccl_count"] == world for j in job), job
    assert all(j["pci_bus_id"] == me["pci_bus_id"] for j in job), job  # (the ranks of this test share GPU 0)
    comm.barrier()
    comm.close()
    print("RANK %d OK %d intervals, shards %s" % (rank, n_iv, sizes), flush=True)


if __name__ == "__main__":
    main()
