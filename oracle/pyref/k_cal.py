#!/usr/bin/env python3
"""DEV-CONTAINER TOOLING (test infrastructure): k_cal = restatement / genuine reference.

BASELINE.md section 4, step 1: the genuine reference (Python + Cython, cannot travel to the GPU box)
and this repo's C restatement (oracle/fpt_oracle.c, what bench.py times there as `cpu_baseline`)
run the SAME BASELINE config-1 workload here -- 1,000 x 500 bp, hw=5, shw=50, clip=0.01,
DM-SYNTH-A, Stouffer hw=3, inputs generated beforehand -- on 1 and on 8 workers.  The ratio lets a
bench line state the implied speed-up over the genuine reference:
    reference bases/s on the GPU box's host  ~=  cpu_baseline / k_cal.

    python oracle/pyref/build_pyref.py && make -C oracle && python oracle/pyref/k_cal.py
writes profiles/k_cal.json.
"""
import json
import multiprocessing as mp
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

N_IV, L, HW, SHW, CLIP, SCALE = 1000, 500, 5, 50, 0.01, 3
LP = L + 2 * (HW + SHW) + 1


def inputs(lo, hi):
    from oracle import oracle
    cp = oracle.synth_counts(1, lo * LP, (hi - lo) * LP, 0)
    cm = oracle.synth_counts(1, lo * LP, (hi - lo) * LP, 1)
    sq = oracle.synth_bases(1, lo * (LP + 6), (hi - lo) * (LP + 6))
    return cp, cm, sq


class _Reads(object):
    def __init__(self, plus, minus):
        self.p, self.m = plus, minus

    def __getitem__(self, iv):
        return {"+": self.p, "-": self.m}


class _Fasta(object):
    def __init__(self, seq):
        self.seq = seq

    def fetch(self, chrom, start, end):
        return self.seq


def reference_worker(args):
    """cli/detect.py:120-130 per interval, with the reference's own modules."""
    lo, hi = args
    import ref_import
    ref_import.load()
    from footprint_tools.modeling import bias, dispersion, predict
    from footprint_tools.stats import windowing
    from oracle import oracle
    bm = bias.kmer_model(os.path.join(ref_import.REF, "data", "vierstra_et_al.6mer-model.txt"))
    dm = dispersion.dispersion_model()
    dm.mu_params, dm.r_params = list(oracle.DM_SYNTH_A["mu"]), list(oracle.DM_SYNTH_A["r"])
    cp, cm, sq = inputs(lo, hi)
    seqs = [sq[i * (LP + 6):(i + 1) * (LP + 6)].tobytes().decode() for i in range(hi - lo)]
    t0 = time.perf_counter()
    acc = 0.0
    for i in range(hi - lo):
        pr = predict.prediction(_Reads(cp[i * LP:(i + 1) * LP], cm[i * LP:(i + 1) * LP]), _Fasta(seqs[i]), bm,
                                half_win_width=HW, smoothing_half_win_width=SHW, smoothing_clip=CLIP)
        obs, exp, _ = pr.compute(ref_import.genomic_interval("chr1", 1000, 1000 + L))
        obs = obs["+"][1:] + obs["-"][:-1]
        exp = exp["+"][1:] + exp["-"][:-1]
        p = np.asarray(dm.p_values(exp, obs))
        wp = windowing.stouffers_z(np.ascontiguousarray(p), SCALE)
        acc += float(np.nansum(wp))
    return time.perf_counter() - t0, acc


def restatement(n_threads):
    from oracle import oracle
    g = np.load(os.path.join(ROOT, "tests", "golden", "kmer_probs.npz"))
    cp, cm, sq = inputs(0, N_IV)
    t0 = time.perf_counter()
    e, o, p, wp = oracle.detect_batch(cp, cm, sq, N_IV, L, HW, SHW, CLIP, g["table"], oracle.DM_SYNTH_A["mu"],
                                      oracle.DM_SYNTH_A["r"], (SCALE,), n_threads=n_threads)
    return time.perf_counter() - t0, float(np.nansum(wp))


def main():
    total = N_IV * L
    out = {"workload": "BASELINE config 1: %d x %d bp, hw=%d shw=%d clip=%g, DM-SYNTH-A, Stouffer hw=%d"
                       % (N_IV, L, HW, SHW, CLIP, SCALE),
           "host": platform.processor() or platform.machine(), "cpu_count": os.cpu_count()}
    try:
        out["cpu_model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    for workers in (1, 8):
        per = N_IV // workers
        chunks = [(w * per, (w + 1) * per if w < workers - 1 else N_IV) for w in range(workers)]
        t0 = time.perf_counter()
        if workers == 1:
            res = [reference_worker(chunks[0])]
            wall = res[0][0]
        else:
            with mp.get_context("fork").Pool(workers) as pool:  # one process per core, like batch_iter(num_workers=8)
                res = pool.map(reference_worker, chunks)
            wall = max(r[0] for r in res)  # compute only (imports and input generation excluded, as for 1 worker)
        ref_rate = total / wall
        ref_sum = sum(r[1] for r in res)
        dt = min(restatement(workers)[0] for _ in range(3))
        rs_sum = restatement(workers)[1]
        assert abs(ref_sum - rs_sum) <= 1e-9 * abs(ref_sum), (ref_sum, rs_sum)  # same results
        out["workers_%d" % workers] = {"reference_bases_per_s": ref_rate, "restatement_bases_per_s": total / dt,
                                       "k_cal": (total / dt) / ref_rate}
        print(workers, out["workers_%d" % workers], flush=True)
    with open(os.path.join(ROOT, "profiles", "k_cal.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
