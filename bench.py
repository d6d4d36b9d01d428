#!/usr/bin/env python3
"""bench.py -- throughput of the fused per-nucleotide footprint scan on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3|2|4|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    (the launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT: bench.py imports no
    torch; the collective is RCCL bound by the library itself, footprint_tools_amd/distributed.py.
    With --gpus N > 1 and NO launcher around it, bench.py starts its N ranks itself: launch_ranks)

One step = one pass of the hot path (6-mer lookup -> expected cleavage with trimmed-mean
smoothing -> NB p-value -> Stouffer windows) over one batch of synthetic intervals that is
already resident in HBM.  N=1 workload = BASELINE.json configs[2], the largest single-GPU
configuration (1,000,000 x 1 kb, five Stouffer scales); `--config 2` selects configs[1]
(100,000 x 500 bp, one scale), `--config 4` the ragged whole-genome set (437,500 intervals per
GPU, ONE global list cut by scan.shard_intervals, ragged all-gather).  With N>1 every rank
scans its own shard of N x batch intervals (weak scaling, no data-path collective inside the
scan) and, after the K steps and still inside the timed region, the per-base p-value track of
the resident batch is re-assembled on every rank with ONE RCCL all-gather (N x 400 MB for
config 2): "a single all-gather at the end", as BASELINE.json's north_star words it.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the fused
kernel and `cpu_baseline` = the C oracle timed on this box's host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1] / configs[2]
    "2": dict(name="100000x500bp_1scale", n_iv=100000, L=500, scales=(3,)),
    "3": dict(name="1000000x1kb_5scales", n_iv=1000000, L=1000, scales=(3, 5, 10, 20, 40)),
    # BASELINE.json configs[3] shape, scaled to one GPU's share (1/8 of ~3.5M intervals):
    # variable-length intervals, lognormal lengths clipped to [50, 2000], mean ~171 bp
    "4": dict(name="437500xragged171bp_1scale", n_iv=437500, L=0, scales=(3,)),
    # BASELINE.json configs[4]: the same set through the full `detect` statistics -- per-interval NB
    # dispersion models (4 models, chosen per interval) + empirical FDR with 100 null draws per base
    # -- and, beside the headline, the posterior caller over 8 datasets of that shape (one launch)
    "5": dict(name="437500xragged171bp_4models_fdr100", n_iv=437500, L=0, scales=(3,), fdr_times=100, n_models=4,
              posterior_datasets=8),
    # small shapes for quick checks
    "1": dict(name="1000x500bp_5scales", n_iv=1000, L=500, scales=(3, 5, 10, 20, 40)),
}
HW, SHW, CLIP = 5, 50, 0.01
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_base(L, n_scales):
    """read (l/L)*(8+8+1) for counts+/-, sequence; write 8*(3+S)."""
    l = L + 2 * (HW + SHW) + 1
    return (l / L) * 17.0, 8.0 * (3 + n_scales)


def load_models():
    g = np.load(os.path.join(ROOT, "tests", "golden", "kmer_probs.npz"))
    lat = np.load(os.path.join(ROOT, "tests", "golden", "nb_lattice.npz"))

    class DM(object):  # DM-SYNTH-A
        mu_params, r_params = lat["mu_A"], lat["r_A"]

    return g["table"], DM


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """CPUs this process can actually run on: the machine's count cut to the affinity mask and to
    the control group's quota (a GPU box reports 256 hardware threads under a quota of 16 CPUs;
    256 OpenMP threads there are 16 cores' worth of work, throttled)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def traffic_probe(argv_cfg, timeout_s=100):
    """HBM bytes per step of the scan kernels and the shader clock they ran at, read from the PMC counters
    in THIS invocation: three child runs of this script (1 warm-up + 2 steps, nothing but the headline leg)
    under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and `--pmc GRBM_GUI_ACTIVE` -- separate passes,
    counters only, as MI355X_MICROARCH.md's HBM section prescribes (both sizes count KiB; on gfx950
    FETCH_SIZE reports half the bytes of a coalesced streaming read and is doubled; GRBM_GUI_ACTIVE is
    summed over the 8 XCDs).  The first dispatch of every kernel (the warm-up call: empty second-level
    table, cold caches) is dropped, the rest averaged.  Started before this process touches the GPU, each
    child in its own process group (killed as a group on timeout).  None when rocprofv3 or the counters
    are not to be had: the line then falls back to profiles/traffic.json (the same measurement, made
    earlier)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    under_profiler = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if not prof or under_profiler:  # (a run that is itself being profiled does not start profilers)
        return None
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    got, clock, issue = {}, None, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
        out = tempfile.mkdtemp(prefix="fpt_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            # (the clock pass also reads how busy the vector and LDS pipes were: same block of counters, same run)
            pmc = [counter] if counter != "GRBM_GUI_ACTIVE" else [counter, "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE"]
            cmd = [prof, "--pmc"] + pmc + ["--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__)] + argv_cfg + ["--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                                                            "--no-other-mode", "--no-heavy", "--no-posterior",
                                                            "--no-traffic-probe", "--no-box-stream", "--no-config-legs",
                                                            "--no-issue-probe", "--no-host-arrays"]
            child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=env["TMPDIR"], env=env,
                                     start_new_session=True)
            try:
                if child.wait(timeout=timeout_s) != 0:
                    raise subprocess.SubprocessError("rocprofv3 child failed")
            except subprocess.TimeoutExpired:
                os.killpg(child.pid, signal.SIGKILL)  # the profiler AND the program under it
                child.wait()
                raise
            per, spans, extra = {}, {}, {}
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] in ("SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE") and "scan_lean" in name:
                        extra[row["Counter_Name"]] = extra.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if row["Counter_Name"] == counter and ("scan_lean" in name or "scan_fused" in name):
                        key = (name, int(row.get("Dispatch_Id", 0) or 0))
                        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
                        if "End_Timestamp" in row and "Start_Timestamp" in row:
                            spans[key] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if not per:
                return None
            by_kernel = {}
            for (name, disp) in sorted(per):
                by_kernel.setdefault(name, []).append((per[(name, disp)], spans.get((name, disp), 0)))
            total = cyc = ns = 0.0
            for name, v in by_kernel.items():
                v = v[len(v) // 3:] if len(v) >= 3 else v  # drop the warm-up step's dispatch(es)
                total += sum(x[0] for x in v) / len(v)
                if "scan_fused" not in name:
                    cyc += sum(x[0] for x in v)
                    ns += sum(x[1] for x in v)
            got[counter] = total
            if counter == "GRBM_GUI_ACTIVE" and ns > 0:
                clock = cyc / 8.0 / ns  # GHz: cycles summed over the 8 XCDs / duration of the first-pass kernels
                all_cyc = sum(per[k_] for k_ in per if "scan_lean" in k_[0]) / 8.0
                if all_cyc > 0 and extra:  # rocprofv3's VALUBusy / LdsUtil of the first-pass kernels (all their dispatches)
                    issue = dict(valu_busy=extra.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * all_cyc),
                                 lds_busy=extra.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * all_cyc))
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            if counter == "GRBM_GUI_ACTIVE" and "FETCH_SIZE" in got and "WRITE_SIZE" in got:
                break  # the traffic stands without the clock
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    rd, wr = 2.0 * got["FETCH_SIZE"] * 1024.0, got["WRITE_SIZE"] * 1024.0
    return dict(read_bytes=rd, write_bytes=wr, bytes_per_launch=rd + wr, shader_clock_GHz=clock, issue=issue,
                raw_counters=dict(FETCH_SIZE_KiB=got.get("FETCH_SIZE"), WRITE_SIZE_KiB=got.get("WRITE_SIZE"),
                                  GRBM_GUI_ACTIVE=got.get("GRBM_GUI_ACTIVE"),
                                  corrections="FETCH_SIZE x2 (gfx950 streaming reads), KiB -> bytes"))


def issue_probe(argv_cfg, timeout_s=150):
    """Config 5's dominant kernels are the null draws of the empirical-FDR pass, bound by the issue of vector
    and LDS instructions, not by HBM: their busy fractions are read from the SQ counters in THIS invocation --
    one child run of this script (1 warm-up + 2 steps) under `rocprofv3 --pmc` (counters only), per group of
    kernels:  VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1,024 SIMDs x cycles)   [rocprofv3's VALUBusy],
              LDS busy  = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)              [LdsUtil],
    cycles = GRBM_GUI_ACTIVE / 8 XCDs summed over the group's dispatches.  None when rocprofv3 is not to be had."""
    import csv
    import glob
    import re
    import shutil
    import signal
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    under_profiler = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if not prof or under_profiler:
        return None
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    counters = ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"]
    # (a pass of its own for the texture addresser -- the sampler's table gathers go through it: counters of
    # another block, and a pass that fails takes only its own figure with it)
    ta_counters = ["TA_TA_BUSY_sum", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"]
    out = tempfile.mkdtemp(prefix="fpt_pmc_", dir=env["TMPDIR"])
    out_ta = tempfile.mkdtemp(prefix="fpt_pmc_ta_", dir=env["TMPDIR"])
    calls = 3

    def counter_run(ctrs, where, limit):
        cmd = [prof, "--pmc"] + ctrs + ["--output-format", "csv", "-d", where, "--", sys.executable, os.path.abspath(__file__)] + \
            argv_cfg + ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-mode", "--no-heavy", "--no-posterior",
                        "--no-traffic-probe", "--no-box-stream", "--no-config-legs", "--no-issue-probe", "--no-host-arrays"]
        child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=env["TMPDIR"], env=env,
                                 start_new_session=True)
        try:
            return child.wait(timeout=limit) == 0
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)
            child.wait()
            return False
    try:
        if not counter_run(counters, out, timeout_s):
            return None
        have_ta = counter_run(ta_counters, out_ta, 60)

        def group_of(name):
            m = re.search(r"k_fdr_null<\d+, \w+, \d+, \w+, (\d+)>", name)
            if m:
                return {"1": "setup", "3": "draws", "2": "full_draws"}.get(m.group(1), "other")
            if "k_fdr_slice_finish" in name:
                return "other"
            if "k_fdr_slice<" in name:
                return "draws"
            if "k_nb_alias" in name:
                return "other"
            return None
        def collect(where):
            agg = {}
            for f in glob.glob(os.path.join(where, "**", "*_counter_collection.csv"), recursive=True):
                seen = set()
                for row in csv.DictReader(open(f)):
                    g = group_of(row["Kernel_Name"])
                    if g is None:
                        continue
                    a = agg.setdefault(g, dict(ns=0.0, dispatches=0))
                    a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    key = row.get("Dispatch_Id")
                    if key not in seen:
                        seen.add(key)
                        a["dispatches"] += 1
                        if "End_Timestamp" in row and "Start_Timestamp" in row:
                            a["ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            return agg
        agg = collect(out)
        agg_ta = collect(out_ta) if have_ta else {}
        if "draws" not in agg or not agg["draws"].get("GRBM_GUI_ACTIVE"):
            return None
        res = {}
        for g, a in agg.items():
            cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            if cyc <= 0:
                continue
            idx = a.get("SQ_LDS_IDX_ACTIVE", 0.0)
            res[g] = dict(valu_busy=a.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * cyc),
                          lds_busy=idx / (256.0 * cyc),
                          lds_bank_conflict_share=(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else None),
                          lds_bank_conflict_cycles_per_lds_instruction=(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_INSTS_LDS"]
                                                                        if a.get("SQ_INSTS_LDS") else None),
                          valu_instructions_per_call=a.get("SQ_INSTS_VALU", 0.0) / calls,
                          lds_instructions_per_call=a.get("SQ_INSTS_LDS", 0.0) / calls,
                          lds_bank_conflict_cycles_per_call=a.get("SQ_LDS_BANK_CONFLICT", 0.0) / calls,
                          ms_per_call_under_profiler=a["ns"] / calls * 1e-6, dispatches_per_call=a["dispatches"] / calls,
                          shader_clock_GHz=(cyc / a["ns"] if a["ns"] else None))
            # texture addresser (one per CU): busy cycles / (256 x cycles of ITS run), and per vector-memory read instruction
            t = agg_ta.get(g, {})
            cyc_ta = t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            res[g]["ta_busy"] = t.get("TA_TA_BUSY_sum", 0.0) / (256.0 * cyc_ta) if cyc_ta > 0 else None
            res[g]["ta_busy_cycles_per_vmem_read_instruction"] = (t.get("TA_TA_BUSY_sum", 0.0) / t["SQ_INSTS_VMEM_RD"]
                                                                  if t.get("SQ_INSTS_VMEM_RD") else None)
        return res
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)
        shutil.rmtree(out_ta, ignore_errors=True)


def run_config_legs(steps_by_cfg, timeout_s=170, extra_args=()):
    """The other BASELINE configurations beside the headline, each a child run of this script (started
    before this process touches the GPU, one after the other): its own buffers, its own parity spot check,
    its own roofline block for ITS dominant kernel.  Returns {config: summary}."""
    import signal
    import subprocess
    legs = {}
    for cid, steps in steps_by_cfg:
        # (a 1 - 2 ms step rides on the clock's ramp out of idle -- the same leg gave 1.8 to 2.5 ms with ten untimed
        # steps: a quarter of a second of them first; config 5's 30 ms steps need two)
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cid, "--steps", str(steps), "--warmup", "2" if cid == "5" else "150",
               "--no-cpu-baseline", "--no-other-mode", "--no-heavy", "--no-config-legs"] + list(extra_args)
        t0 = time.perf_counter()
        try:
            child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, start_new_session=True)
            try:
                o, e = child.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(child.pid, signal.SIGKILL)
                child.communicate()
                legs[cid] = dict(error="timed out after %d s" % timeout_s)
                continue
            lines = [ln for ln in o.decode().splitlines() if ln.startswith("{")]
            if child.returncode != 0 or not lines:
                legs[cid] = dict(error="rc %d: %s" % (child.returncode, e.decode()[-400:]))
                continue
            d = json.loads(lines[-1])
            legs[cid] = leg_summary(d, time.perf_counter() - t0)
        except (OSError, ValueError) as ex:
            legs[cid] = dict(error=str(ex))
    return legs


def leg_summary(d, wall_s=None):
    """what a configuration's entry of `configs` carries"""
    out = dict(workload=d["config"]["workload"], value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"],
               steps=d["steps"], warmup=d["warmup"], dtype=d["dtype"], parity=d["parity"], roofline=d["roofline"])
    if d.get("posterior"):
        out["posterior"] = d["posterior"]
    if d.get("fdr"):
        out["fdr"] = d["fdr"]
    if d.get("degraded"):
        out["degraded"] = d["degraded"]
    if wall_s is not None:
        out["leg_wall_s"] = wall_s
    return out


def cpu_baseline(cfg, table, DM, budget_s=12.0):
    """The CPU oracle (a port of the reference algorithm, results identical to the reference on
    the golden vectors) on a bounded sample of the same workload: all host cores, and one core.
    k_cal (profiles/k_cal.json, measured in the build container with oracle/pyref/k_cal.py) is
    the ratio port / genuine reference on BASELINE config 1, so value / k_cal estimates what the
    reference itself (Python + Cython, cannot travel here) would do on these cores."""
    from oracle import oracle  # checker / baseline only
    L, scales = cfg["L"], cfg["scales"]
    l = L + 2 * (HW + SHW) + 1
    cores = usable_cpus()

    def run(n, threads):
        cp = oracle.synth_counts(1, 0, n * l, 0)
        cm = oracle.synth_counts(1, 0, n * l, 1)
        sq = oracle.synth_bases(1, 0, n * (l + 6))
        t0 = time.perf_counter()
        oracle.detect_batch(cp, cm, sq, n, L, HW, SHW, CLIP, table, DM.mu_params, DM.r_params, scales,
                            n_threads=threads)
        return time.perf_counter() - t0

    probe_n = 4 * cores
    dt = run(probe_n, cores)
    rate = probe_n * L / dt
    n = int(max(probe_n, min(rate * budget_s / L, 200000)))
    dt = run(n, cores)
    if dt < 0.5 * budget_s and n < 200000:  # the probe was mostly thread start-up: size the sample again
        n = int(max(n, min(n * budget_s / dt, 200000)))
        dt = run(n, cores)
    dt1 = run(8, 1)
    n1 = int(max(8, min(8 * 3.0 / dt1, 20000)))  # about 3 s on one core
    dt1 = run(n1, 1)
    out = dict(value=n * L / dt, unit="bases/s", cores=cores, kind="port", cpu_model=cpu_model(),
               value_1core=n1 * L / dt1,
               sample="%d intervals x %d bp x %d scales of the same synthetic workload, oracle/fpt_oracle.c with "
                      "OpenMP over intervals on %d threads, %.1f s; 1-core figure on %d intervals, %.1f s"
                      % (n, L, len(scales), cores, dt, n1, dt1))
    out["reference_native"] = reference_native(cfg, table, DM, cores)
    # the same figures as plain scalars of this block (nested dicts do not survive every reader of the line)
    rn = out["reference_native"]
    out["reference_native_1core"] = rn["value_1core"] if rn else None
    out["reference_native_allcores"] = rn["value"] if rn else None
    kc = os.path.join(ROOT, "profiles", "k_cal.json")
    if os.path.exists(kc):
        k = json.load(open(kc))
        # k_cal = port / whole reference (its Python + Cython loops around the native code included), measured
        # in the BUILD container on another CPU: it only prices the interpreter overhead the native figure
        # above leaves out, and is quoted with the CPU it came from
        out["k_cal"] = dict(port_over_reference_1_worker=k["workers_1"]["k_cal"],
                            port_over_reference_8_workers=k["workers_8"]["k_cal"],
                            measured_on=k.get("cpu_model"), workload=k["workload"],
                            note="measured in the build container, not on this box: covers the reference's Python-loop "
                                 "overhead only; its native C is timed on this box in reference_native")
        out["implied_reference_1core"] = out["value_1core"] / k["workers_1"]["k_cal"]
        out["k_cal_1worker"] = k["workers_1"]["k_cal"]
        out["k_cal_8workers"] = k["workers_8"]["k_cal"]
        out["k_cal_cpu"] = k.get("cpu_model")
    return out


def reference_native(cfg, table, DM, cores, budget_s=3.0):
    """The reference's OWN native code -- oracle/_ref/libfpt_ref.so: fast_predict (modeling/predict.h:23-74),
    hcephes_incbet (hcephes/src/cprob/incbet.c:12-94) and fast_windowing_func + fast_stouffers_z
    (stats/windowing.h:53-84), compiled from /root/reference in the build container by oracle/Makefile --
    timed on this box on a sample of the same workload, on one core and on all usable cores (threads:
    ctypes releases the interpreter lock around the calls).  What the reference does in Python / Cython
    around these calls (6-mer lookup, fit_mu / fit_r per base, the strand merge) is prepared OUTSIDE the
    timed loop.  None when the library did not travel."""
    from oracle import oracle  # checker / baseline only
    R = oracle.ref_lib()
    if R is None:
        return None
    from concurrent.futures import ThreadPoolExecutor
    L, scales = cfg["L"], cfg["scales"]
    pad = HW + SHW
    l = L + 2 * pad + 1
    mu_par, r_par = np.asarray(DM.mu_params, np.float64), np.asarray(DM.r_params, np.float64)

    def prepare(n):
        cp = oracle.synth_counts(1, 0, n * l, 0).reshape(n, l)
        cm = oracle.synth_counts(1, 0, n * l, 1).reshape(n, l)
        sq = oracle.synth_bases(1, 0, n * (l + 6)).reshape(n, l + 6)
        items = []
        for i in range(n):
            fwd, rev = oracle.kmer_probs(sq[i], table)[:2]
            # exp / obs of the merged strands and the NB arguments per base (Python in the reference)
            e, o, _, _ = oracle.detect_batch(cp[i], cm[i], sq[i], 1, L, HW, SHW, CLIP, table, mu_par, r_par, scales[:1])
            r = np.array([oracle.fit_r(r_par, x) for x in e])
            mu = np.array([oracle.fit_mu(mu_par, x) for x in e])
            items.append((cp[i].copy(), cm[i].copy(), np.ascontiguousarray(fwd), np.ascontiguousarray(rev),
                          np.ascontiguousarray(r), np.floor(o) + 1.0, np.ascontiguousarray(r / (r + mu))))
        return items

    def native(item):
        cp, cm, fwd, rev, a, b, x = item
        e, w = np.empty(l), np.empty(l)
        R.ref_fast_predict(cp, fwd, l, HW, SHW, CLIP, e, w)
        R.ref_fast_predict(cm, rev, l, HW, SHW, CLIP, e, w)
        p = np.empty(L)
        R.ref_incbet_v(a, b, x, L, p)
        out = np.empty(L)
        for hw in scales:
            R.ref_window(3, p, None, L, int(hw), out)

    items = prepare(8)
    t0 = time.perf_counter()
    for it in items:
        native(it)
    per = (time.perf_counter() - t0) / len(items)
    n1 = int(max(8, min(budget_s / per, 4000)))
    items = prepare(n1) if n1 > len(items) else items
    t0 = time.perf_counter()
    for it in items:
        native(it)
    dt1 = time.perf_counter() - t0
    # all cores: the same items round and round until the budget is used, `cores` threads
    reps = max(1, int(cores * budget_s / max(dt1, 1e-9)))
    work = items * reps
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(native, work, chunksize=max(1, len(work) // (cores * 8))))
    dtn = time.perf_counter() - t0
    return dict(kind="reference", value=len(work) * L / dtn, value_1core=len(items) * L / dt1, unit="bases/s", cores=cores,
                functions="fast_predict x 2 strands (modeling/predict.h:23-74), hcephes_incbet per base "
                          "(hcephes/src/cprob/incbet.c:12-94), fast_windowing_func + fast_stouffers_z per scale "
                          "(stats/windowing.h:53-84): oracle/_ref/libfpt_ref.so, gcc -O2 -fwrapv as the reference's build",
                sample="%d intervals x %d bp x %d scales on 1 core in %.1f s; %d on %d threads in %.1f s; the reference's "
                       "Python-side steps (6-mer lookup, fit_mu / fit_r, strand merge) prepared outside the timed loop"
                       % (len(items), L, len(scales), dt1, len(work), cores, dtn))


def host_arrays_leg(ctx, table, DM):
    """The rate a drop-in Python caller sees: numpy arrays in, numpy arrays out through FootprintScanner.scan
    (fpt_scan_host: chunks through a copy-in / scan / copy-out pipeline), BASELINE config 2's size, PCIe included --
    with pageable arrays (np.empty: pinned by the runtime on the way, every copy blocking the thread that issued it --
    hence the call's two threads) and with page-locked ones (ctx.pinned_empty).  Reference counterpart: the per-call API
    modeling/predict.pyx:116-163 + dispersion.pyx:291-316 + windowing.pyx:114-130 (arrays in, arrays out).
    Never `value`: the headline has its inputs resident in HBM."""
    from oracle import oracle  # inputs (the synthetic generator) and the spot check only
    from footprint_tools_amd.scan import FootprintScanner
    cfg = CONFIGS["2"]
    n_iv, L, scales = cfg["n_iv"], cfg["L"], cfg["scales"]
    sc = FootprintScanner(table, DM, HW, SHW, CLIP, scales, ctx=ctx, nb_mode="memo")
    l = sc.padded_len(L)
    cp, cm = oracle.synth_counts(1, 0, n_iv * l, 0), oracle.synth_counts(1, 0, n_iv * l, 1)
    sq = oracle.synth_bases(1, 0, n_iv * (l + 6))
    out = {}
    res = {}
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            t0 = time.perf_counter()
            a_in = [ctx.pinned_empty(a.shape, a.dtype) for a in (cp, cm, sq)]
            for dst, src in zip(a_in, (cp, cm, sq)):
                dst[...] = src
            alloc_s = time.perf_counter() - t0
        else:
            a_in, alloc_s = (cp, cm, sq), 0.0
        chunk = int(os.environ.get("FPT_BENCH_HOST_CHUNK", "0"))
        # the first call makes the output arrays (and brings the pipeline's buffers up); the timed calls write
        # them again (out=): what a caller that loops over batches does
        t0 = time.perf_counter()
        r = sc.scan(a_in[0], a_in[1], a_in[2], interval_len=L, pinned_out=kind == "pinned", chunk_bases=chunk)
        first_s = time.perf_counter() - t0
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r = sc.scan(a_in[0], a_in[1], a_in[2], interval_len=L, chunk_bases=chunk, out=r)
            dt = time.perf_counter() - t0
            st = ctx.scan_host_last()
            if best is None or dt < best[0]:
                best = (dt, st)
        fresh_s = None
        if kind == "pageable":  # ... and one call that makes NEW output arrays again, the pipeline's buffers warm (what a
            t0 = time.perf_counter()  # caller that does not hand `out` back pays: the pages are touched by a team first)
            r2 = sc.scan(a_in[0], a_in[1], a_in[2], interval_len=L, chunk_bases=chunk)
            fresh_s = time.perf_counter() - t0
            del r2
        res[kind] = r
        dt, st = best
        out[kind] = dict(value=n_iv * L / dt, unit="bases/s", ms_per_call=dt * 1e3, pipeline_ms=st["seconds"] * 1e3,
                         first_call_ms=first_s * 1e3, chunks=st["chunks"],
                         link_GBps_h2d=st["bytes_h2d"] / st["seconds"] / 1e9,
                         link_GBps_d2h=st["bytes_d2h"] / st["seconds"] / 1e9,
                         link_GBps_both=(st["bytes_h2d"] + st["bytes_d2h"]) / st["seconds"] / 1e9,
                         arrays_page_locked=bool(st["inputs_pinned"] and st["outputs_pinned"]),
                         calling_thread_ms=dict(waiting=st["wait_seconds"] * 1e3, issuing=st["issue_seconds"] * 1e3))
        if fresh_s is not None:
            out[kind]["new_output_arrays_ms"] = fresh_s * 1e3
            out[kind]["new_output_arrays_value"] = n_iv * L / fresh_s
        if kind == "pinned":
            out[kind]["input_alloc_and_fill_s"] = alloc_s
    iv = n_iv - 1
    e, o, p, wp = oracle.detect_batch(cp[iv * l:(iv + 1) * l], cm[iv * l:(iv + 1) * l], sq[iv * (l + 6):(iv + 1) * (l + 6)], 1, L,
                                      HW, SHW, CLIP, table, DM.mu_params, DM.r_params, scales)
    r = res["pageable"]
    sl = slice(iv * L, (iv + 1) * L)
    same = all(np.array_equal(res["pageable"][k], res["pinned"][k], equal_nan=True) for k in ("exp", "obs", "pval", "winp"))
    out["parity"] = dict(exp_bit_exact=bool(np.array_equal(r["exp"][sl], e) and np.array_equal(r["obs"][sl], o)),
                         p_max_rel_err=float(np.nanmax(np.abs(r["pval"][sl] - p) / np.maximum(np.abs(p), 1e-300))),
                         winp_max_rel_err=float(np.nanmax(np.abs(r["winp"][:, sl] - wp) / np.maximum(np.abs(wp), 1e-300))),
                         pinned_equals_pageable_bitwise=bool(same))
    out["workload"] = (cfg["name"] + ", numpy arrays in and out of FootprintScanner.scan (PCIe-inclusive; best of 3 calls "
                       "that write the output arrays of a first call again: first_call_ms is the one that made them)")
    out["bytes_per_base_over_the_link"] = dict(h2d=(l / L) * 17.0, d2h=8.0 * (3 + len(scales)))
    return out


def launch_ranks(n, argv, timeout_s=None, program=None):
    """`python3 bench.py --gpus N` with no launcher around it: this process starts the N ranks itself -- the
    counterpart of the reference's `batch_iter(num_workers=n)` forking its own workers (cli/detect.py:394).
    Called before anything here has touched the GPU.  Every rank is a fresh `python bench.py <same argv>` in its
    own session with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run
    would set; the externally launched form keeps working); rank 0's stdout -- the one JSON line -- is this
    process's stdout, the other ranks' stdout is dropped, every rank's stderr is this one's.  If a rank fails,
    or the job outlives FPT_LAUNCH_TIMEOUT_S (default 1800), every rank's process group is killed and the exit
    code is non-zero.  Returns the exit code.  (`program`: tests start another script in place of this file.)"""
    import signal
    import socket
    import subprocess
    if timeout_s is None:
        timeout_s = float(os.environ.get("FPT_LAUNCH_TIMEOUT_S", "1800"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # the communicator id travels through a file named after this job alone (distributed._id_path)
    comm_file = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fpt_comm_bench_%d_%d.id" % (os.getpid(), port))
    procs = []
    devnull = open(os.devnull, "wb")

    def kill_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGKILL)  # (start_new_session: pgid == pid)
                except (ProcessLookupError, PermissionError):
                    pass
        for q in procs:
            try:
                q.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass

    def on_signal(signum, _frame):
        kill_all()
        os._exit(128 + signum)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.setdefault("FPT_COMM_FILE", comm_file)
            procs.append(subprocess.Popen([sys.executable, program or os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=None if r == 0 else devnull, start_new_session=True))
        t_end = time.time() + timeout_s
        live = set(range(n))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is None:
                    continue
                live.discard(r)
                if c != 0:
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other %d ranks\n" % (r, c, len(live)))
                    rc = c if 0 < c < 256 else 1
                    live.clear()
                    break
            if live and time.time() > t_end:
                sys.stderr.write("bench.py: ranks %s still running after %.0f s (FPT_LAUNCH_TIMEOUT_S); stopping the job\n"
                                 % (sorted(live), timeout_s))
                rc = 124
                break
            if live:
                time.sleep(0.05)
    finally:
        kill_all()
        for sg, h in old.items():
            signal.signal(sg, h)
        devnull.close()
        try:
            os.unlink(comm_file)
        except OSError:
            pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="3", choices=sorted(CONFIGS))
    ap.add_argument("--intervals", type=int, default=0, help="override the number of intervals per GPU")
    ap.add_argument("--scales", default=None, help="override the Stouffer half-widths, comma-separated ('' = none): diagnostics")
    ap.add_argument("--nb-mode", default="memo", choices=["memo", "direct"],
                    help="per-base NB p-value: exact (exp,obs) memo table rebuilt inside every step, "
                         "or direct incbet per base; at N=1 the other mode is timed too and reported")
    ap.add_argument("--hotspots", type=int, default=0, metavar="PER_MILLE",
                    help="heavy-tailed variant of the workload for the headline run: this share of the "
                         "intervals carries a hotspot burst (observed counts up to ~1000)")
    ap.add_argument("--no-heavy", action="store_true",
                    help="N=1: skip the extra heavy-tailed measurement (20 per mille hotspots) reported beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-posterior", action="store_true", help="config 5: skip the posterior-caller leg")
    ap.add_argument("--no-traffic-probe", action="store_true",
                    help="N=1: do not read the HBM counters in child runs under rocprofv3 (roofline.traffic then "
                         "comes from profiles/traffic.json)")
    ap.add_argument("--no-other-mode", action="store_true", help="N=1: do not time the other nb mode")
    ap.add_argument("--no-box-stream", action="store_true",
                    help="N=1: do not time the scan's access pattern without arithmetic (roofline.box_stream_GBps)")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="N=1, default config: do not run the other BASELINE configurations (2, 4, 5) as child runs reported "
                         "under `configs`")
    ap.add_argument("--leg-intervals", type=int, default=0,
                    help="tests: run the configuration legs on this many intervals each (and allow them beside --intervals)")
    ap.add_argument("--no-issue-probe", action="store_true",
                    help="config 5: do not read the SQ counters of the FDR kernels in a child run under rocprofv3")
    ap.add_argument("--no-host-arrays", action="store_true",
                    help="N=1, default config: skip the host-arrays leg (numpy in / numpy out through FootprintScanner.scan, "
                         "config 2's size, PCIe included)")
    ap.add_argument("--no-allgather", action="store_true", help="N>1: skip the assembly of the p-value track (same as --assembly none)")
    ap.add_argument("--assembly", default="allgather", choices=["allgather", "gather", "none"],
                    help="N>1: how the per-base track is re-assembled -- every rank gets it (one RCCL all-gather: BASELINE.json's "
                         "wording, the default), rank 0 gets it (grouped send / recv to the rank that writes), or not at all")
    ap.add_argument("--allgather", action="store_true",
                    help="N=1: run the track all-gather anyway, on a one-rank RCCL communicator")
    ap.add_argument("--share-gpu", action="store_true",
                    help="smoke-test mode for boxes with fewer GPUs than ranks: every rank uses GPU 0 "
                         "(RCCL refuses two ranks on one device, so the all-gather is skipped)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # no launcher around this process: be the launcher (this process has made no GPU call yet)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and args.gpus != world:
        sys.exit("bench.py --gpus %d inside a job of WORLD_SIZE=%d: the launcher's --nproc-per-node and --gpus "
                 "must agree" % (args.gpus, world))
    cfg = CONFIGS[args.config]
    if args.intervals:
        cfg = dict(cfg, n_iv=args.intervals, name=cfg["name"].replace(str(cfg["n_iv"]), str(args.intervals), 1))
    if args.scales is not None:
        sc_ = tuple(int(x) for x in args.scales.split(",") if x.strip())
        cfg = dict(cfg, scales=sc_, name=cfg["name"] + "+scales_%s" % "_".join(map(str, sc_)))
    n_iv, L, scales = cfg["n_iv"], cfg["L"], cfg["scales"]
    fdr_times, n_models = cfg.get("fdr_times", 0), cfg.get("n_models", 1)
    S = len(scales)
    table, DM = load_models()

    # the CPU baseline runs first: it is plain host work and must not fork after GPU init
    base = None
    if world == 1 and not args.no_cpu_baseline and cfg["L"] > 0:
        base = cpu_baseline(cfg, table, DM)

    # ... and so do the two counter passes (children under rocprofv3; this process has no GPU state yet)
    live_traffic = None
    if world == 1 and not args.no_traffic_probe and not args.hotspots:
        probe_argv = ["--config", args.config, "--nb-mode", args.nb_mode]
        if args.intervals:
            probe_argv += ["--intervals", str(args.intervals)]
        live_traffic = traffic_probe(probe_argv)

    # ... the FDR kernels' issue counters (config 5) ...
    issue = None
    if world == 1 and fdr_times and not args.no_issue_probe and not args.intervals:
        issue = issue_probe(["--config", args.config, "--nb-mode", args.nb_mode])
    # ... and the other BASELINE configurations, each a child run with its own line (reported under `configs`)
    legs = None
    if (world == 1 and args.config == "3" and not args.no_config_legs and (not args.intervals or args.leg_intervals)
            and not args.hotspots and args.scales is None and args.nb_mode == "memo"):
        extra = []
        if args.leg_intervals:  # (tests: small legs, no counter passes)
            extra = ["--intervals", str(args.leg_intervals), "--no-traffic-probe", "--no-issue-probe"]
        legs = run_config_legs([("2", 100), ("4", 100), ("5", 6)], extra_args=extra)

    # a child run that failed (rocprofv3 missing, a timeout) leaves its block without counters: say so in the line
    degraded = []
    if world == 1 and not args.no_traffic_probe and not args.hotspots and live_traffic is None:
        degraded.append("traffic_probe: no HBM counters from this invocation (roofline.traffic from profiles/traffic.json, or null)")
    if world == 1 and fdr_times and not args.no_issue_probe and not args.intervals and issue is None:
        degraded.append("issue_probe: no SQ / TA counters of the FDR kernels from this invocation")
    for cid, leg in (legs or {}).items():
        if leg is None or "error" in leg:
            degraded.append("config %s leg: %s" % (cid, (leg or {}).get("error", "no result")))

    from footprint_tools_amd import _lib
    from footprint_tools_amd.scan import DeviceArray, FootprintScanner, shard_intervals

    ctx = _lib.Context(0 if args.share_gpu else local_rank)
    # ---- the per-call API a drop-in caller uses: host arrays in and out, PCIe included (never `value`).  Run FIRST,
    #      before the resident batch's ~100 GB are allocated: the same leg behind them (allocated and freed again) moves
    #      8.2e8 bases/s where it moves 1.4e9 in a fresh process -- the pipeline's device buffers then come out of
    #      fragmented memory (gpurun_out/r06_n)
    host_leg = None
    if (world == 1 and args.config == "3" and not args.no_host_arrays and not args.hotspots and args.nb_mode == "memo"
            and args.scales is None):
        try:
            host_leg = host_arrays_leg(ctx, table, DM)
        except Exception as ex:  # (a leg beside the headline: its failure is reported, not fatal)
            host_leg = dict(error="%s: %s" % (type(ex).__name__, ex))
    models = DM
    if n_models > 1:  # per-interval dispersion models: DM-SYNTH-A and variants with scaled 1/r
        lat = np.load(os.path.join(ROOT, "tests", "golden", "nb_lattice.npz"))

        def variant(k):
            r = np.array(lat["r_A"], dtype=np.float64)
            r[5:] *= 1.0 + 0.15 * k  # intercepts and slopes of the 1/r fit
            return type("DMv", (), dict(mu_params=lat["mu_A"], r_params=r))
        models = [variant(k) for k in range(n_models)]
    sc = FootprintScanner(table, models, HW, SHW, CLIP, scales, ctx=ctx, nb_mode=args.nb_mode)
    comm = None
    if args.no_allgather:
        args.assembly = "none"
    # (--share-gpu: ranks on one GPU, which RCCL refuses -- unless FPT_RCCL_LIB binds the test suite's stand-in)
    can_comm = not args.share_gpu or bool(os.environ.get("FPT_RCCL_LIB"))
    do_gather = (world > 1 and args.assembly != "none" and can_comm) or (world == 1 and args.allgather)
    to_root = args.assembly == "gather"  # the track on rank 0 only
    if (world > 1 and can_comm) or do_gather:
        from footprint_tools_amd.distributed import TrackComm
        comm = TrackComm(ctx, rank, world)  # RCCL, bound by the library; no torch
    ragged = L == 0
    pad2 = 2 * (HW + SHW)
    if ragged:
        # ONE global interval list (BASELINE.json configs[3]: the whole-genome DHS set is ~3.5M
        # intervals / ~600 Mb on 8 GPUs; here n_iv intervals per GPU), cut into contiguous ranges
        # balanced by padded bases; the synthetic data is a function of the global position, so
        # the job does not depend on the number of ranks
        rs = np.random.RandomState(4)
        lens_all = np.clip(rs.lognormal(4.9, 0.62, n_iv * world), 50, 2000).astype(np.int64)
        if os.environ.get("FPT_BENCH_ALIGN"):  # diagnostics: every interval a multiple of this many bases (aligned track segments)
            q = int(os.environ["FPT_BENCH_ALIGN"])
            lens_all = np.maximum(q, (lens_all // q) * q)
        bounds = shard_intervals(lens_all, world, HW + SHW)
        a_iv, b_iv = bounds[rank]
        lens = lens_all[a_iv:b_iv]
        n_iv = int(lens.size)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        total = int(off[-1])
        counts = [int(lens_all[a:b].sum()) for a, b in bounds]  # bases per rank
        pos0_counts = int((lens_all[:a_iv] + pad2 + 1).sum())
        pos0_seq = int((lens_all[:a_iv] + pad2 + 7).sum())
        n_counts, n_seq = sc.input_sizes(n_iv, total)
    else:
        l = sc.padded_len(L)
        total = n_iv * L  # bases per rank per step
        counts = [total] * world
        n_counts, n_seq = n_iv * l, n_iv * (l + 6)
    total_all = int(sum(counts))

    # ---- resident buffers: inputs generated on the device, outputs written in place
    d_cp, d_cm = DeviceArray(ctx, n_counts * 8), DeviceArray(ctx, n_counts * 8)
    d_sq = DeviceArray(ctx, n_seq)
    d_out = DeviceArray(ctx, (3 + S) * total * 8)   # exp, obs, winp[S], p (unless it lives in the gathered track)
    # the assembled track, twice (the overlapped leg sends one while the next batch is scanned into the
    # other); a rank that only sends (gather to rank 0) needs its own slice only.  The p-value track of
    # the resident batch is written straight into its slice of the assembled track: the collective runs
    # in place
    holds_all = do_gather and (not to_root or rank == 0)
    my_off = int(sum(counts[:rank])) * 8 if holds_all else 0
    d_gathers = [DeviceArray(ctx, (total_all if holds_all else max(total, 1)) * 8) for _ in range(2)] if do_gather else None
    d_gather = d_gathers[0] if do_gather else None
    p_p = d_gather.ptr + my_off if do_gather else d_out.ptr + (2 + S) * total * 8
    p_cp, p_cm, p_sq, p_out = d_cp.ptr, d_cm.ptr, d_sq.ptr, d_out.ptr
    d_off = d_dm = d_efdr = None
    if ragged:
        _lib.check(ctx.L.fpt_synth_dev(ctx.h, 1, pos0_counts, n_counts, p_cp, p_cm, pos0_seq, n_seq, p_sq))
        d_off = DeviceArray(ctx, off.nbytes).upload(off)
        if n_models > 1:  # model of an interval = hash of its GLOBAL index: independent of the sharding
            ids = ((np.arange(a_iv, b_iv, dtype=np.int64) * 2654435761) >> 7) % n_models
            d_dm = DeviceArray(ctx, max(n_iv, 1) * 4).upload(ids.astype(np.int32))
        if fdr_times:
            d_efdr = DeviceArray(ctx, total * 8) if not do_gather else None
    else:
        # rank r owns intervals [r*n_iv, (r+1)*n_iv) of the global synthetic job
        sc.synth_dev(1, n_iv, L, p_cp, p_cm, p_sq, first_interval=rank * n_iv)
        if args.hotspots:
            sc.synth_hotspots_dev(1, n_iv, L, p_cp, p_cm, args.hotspots, first_interval=rank * n_iv)
    ctx.synchronize()

    t8 = total * 8

    # config 5: the gathered track is the empirical FDR; the p-values go to a plain buffer
    if fdr_times:
        d_pv = DeviceArray(ctx, total * 8)
        p_track = d_gather.ptr + my_off if do_gather else d_efdr.ptr
        p_p = d_pv.ptr
        bases_before = int(lens_all[:a_iv].sum())

    fdr_marks = []  # config 5: (before, after) marks around every fpt_fdr_dev of the timed region

    def step_once(bi=0):
        """one step; bi: which of the two assembled-track buffers takes this step's track"""
        shift = d_gathers[bi].ptr - d_gathers[0].ptr if do_gather else 0
        sc.scan_dev(n_iv, p_cp, p_cm, p_sq, exp_out=p_out, obs_out=p_out + t8,
                    pval_out=p_p + (0 if fdr_times else shift), winp_out=p_out + 2 * t8 if S else None,
                    interval_len=None if ragged else L, interval_off_dev=d_off.ptr if ragged else None,
                    interval_off_host=off if ragged else None, dm_ids_dev=d_dm.ptr if d_dm else None)
        if fdr_times:  # detect.py:132-135; null draws keyed by the GLOBAL base index
            m0 = ctx.mark() if fdr_marks is not None else None
            sc.fdr_dev(n_iv, p_out, p_out + 2 * t8, p_track + shift, times=fdr_times, seed=1, half_win_width=scales[0],
                       interval_off_dev=d_off.ptr, base_index0=bases_before, dm_ids_dev=d_dm.ptr if d_dm else None,
                       obs=p_out + t8, interval_off_host=off)
            if m0 is not None:
                fdr_marks.append((m0, ctx.mark()))

    def sync():
        ctx.synchronize()
        if comm is not None:
            comm.barrier()

    def gather_track(bi=0, overlapped=False):
        """The one collective of the job: the whole track on every rank (all-gather), or on rank 0 (gather);
        `overlapped`: on the communicator's own stream, behind this step's scan and beside the next one"""
        if not do_gather:
            return
        shift = d_gathers[bi].ptr - d_gathers[0].ptr
        send = (p_track if fdr_times else p_p) + shift
        recv = d_gathers[bi].ptr if holds_all else None
        if to_root:
            (comm.gather_dev_async if overlapped else comm.gather_dev)(send, counts, recv, root=0)
        else:
            (comm.allgather_dev_async if overlapped else comm.allgather_dev)(send, counts, recv)

    cold_tables = [False]  # heavy-tailed leg: empty the kept second-level table before every call

    def measure(steps, warmup):
        plain_step = step_once

        def step():
            if cold_tables[0]:
                ctx.drop_kept_tables()
            plain_step()
        for _ in range(warmup):
            step()
        gather_track()  # also brings the communicator's channels up before the timed region
        sync()
        ctx.timing_enable(steps)
        if fdr_marks is not None:
            del fdr_marks[:]
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.synchronize()
        t1 = time.perf_counter()  # scans done on this rank; the collective follows
        gather_track()
        sync()
        t2 = time.perf_counter()
        seq_ms, main_ms = ctx.timing_read()
        return t2 - t0, main_ms, seq_ms, t1 - t0, t2 - t1

    def measure_overlapped(steps):
        """every step's track assembled, on the communicator's stream beside the next step's scan: two
        track buffers in turn, the scan into a buffer waits for the collective that last read it"""
        sync()
        t0 = time.perf_counter()
        for k in range(steps):
            comm.wait(back=1)
            step_once(k % 2)
            gather_track(k % 2, overlapped=True)
        comm.synchronize()
        sync()
        return time.perf_counter() - t0

    dt, kernel_ms, seq_ms, dt_scan, dt_gather = measure(args.steps, args.warmup)
    fdr_ms = [ctx.mark_elapsed(a_, b_) for a_, b_ in fdr_marks[-args.steps:]] if fdr_times else None
    fdr_marks = None  # (later calls of step_once are not part of the measurement)
    ctx.marks_clear()
    dt_asm = measure_overlapped(args.steps) if do_gather else None
    if do_gather:  # leave the resident batch's track assembled in buffer 0 for the checks below
        step_once(0)
        gather_track(0)
        sync()
    other = None
    if world == 1 and not args.no_other_mode:  # the other evaluation mode, reported beside the headline
        main_mode = sc.nb_mode
        sc.nb_mode = _lib.NB_DIRECT if args.nb_mode == "memo" else _lib.NB_MEMO
        k2 = max(3, args.steps // 4)
        dt2, kms2 = measure(k2, 1)[:2]
        other = dict(nb_pvalue="direct incbet per base" if args.nb_mode == "memo" else "memo table",
                     value=total * k2 / dt2, unit="bases/s", ms_per_step=dt2 / k2 * 1e3,
                     kernel_ms=float(np.mean(kms2)), steps=k2)
        sc.nb_mode = main_mode
        step_once()  # leave the headline mode's outputs in the buffers for the parity check
        sync()

    job = comm.job_info() if comm is not None else None  # what RCCL itself says about the job, every rank's
    if comm is not None:  # the contract: the slowest rank's time
        dt = comm.max_over_ranks(dt)
        dt_scan = comm.max_over_ranks(dt_scan)
        dt_gather = comm.max_over_ranks(dt_gather)
        if dt_asm is not None:
            dt_asm = comm.max_over_ranks(dt_asm)

    # ---- parity spot check outside the timed region (oracle = checker only)
    parity = None
    if rank == 0:
        from oracle import oracle
        iv = n_iv - 1
        if ragged:
            Li = int(lens[iv])
            li = Li + pad2 + 1
            c0, s0 = pos0_counts + int(off[iv]) + iv * (pad2 + 1), pos0_seq + int(off[iv]) + iv * (pad2 + 7)
            cp, cm = oracle.synth_counts(1, c0, li, 0), oracle.synth_counts(1, c0, li, 1)
            sq = oracle.synth_bases(1, s0, li + 6)
            o0 = int(off[iv])
        else:
            Li, g0 = L, rank * n_iv + iv
            cp = oracle.synth_counts(1, g0 * l, l, 0)
            cm = oracle.synth_counts(1, g0 * l, l, 1)
            if args.hotspots:
                oracle.synth_hotspots(cp, 1, g0 * l, 0, l, args.hotspots)
                oracle.synth_hotspots(cm, 1, g0 * l, 1, l, args.hotspots)
            sq = oracle.synth_bases(1, g0 * (l + 6), l + 6)
            o0 = iv * L
        m_last = models[int(((a_iv + iv) * 2654435761 >> 7) % n_models)] if n_models > 1 else DM
        e, o, p, wp = oracle.detect_batch(cp, cm, sq, 1, Li, HW, SHW, CLIP, table, m_last.mu_params,
                                          m_last.r_params, scales)
        ge = d_out.download(np.float64, Li, o0 * 8)
        if fdr_times:
            gp = d_pv.download(np.float64, Li, o0 * 8)
        else:
            src = d_gather if do_gather else d_out
            gp = src.download(np.float64, Li, (my_off if do_gather else (2 + S) * total * 8) + o0 * 8)
        rel = float(np.nanmax(np.abs(gp - p) / np.maximum(np.abs(p), 1e-300)))
        parity = dict(exp_bit_exact=bool(np.array_equal(ge, e)), p_max_rel_err=rel)
        if S:  # the window p-values of every scale (contract: 1e-6 relative)
            wrel = 0.0
            for s_ in range(S):
                gw = d_out.download(np.float64, Li, ((2 + s_) * total + o0) * 8)
                wrel = max(wrel, float(np.nanmax(np.abs(gw - wp[s_]) / np.maximum(np.abs(wp[s_]), 1e-300))))
            parity["winp_max_rel_err"] = wrel
        if fdr_times:  # the empirical FDR of that interval against the oracle's restatement of the sampler
            ef_src = d_gather if do_gather else d_efdr
            ef = ef_src.download(np.float64, Li, (my_off if do_gather else 0) + o0 * 8)
            ef_want = oracle.fdr_null(m_last.mu_params, m_last.r_params, e, wp[0], scales[0], fdr_times, seed=1,
                                      base0=bases_before + o0)
            parity["efdr_max_abs_err"] = float(np.max(np.abs(ef - ef_want)))
        if do_gather and world > 1 and not fdr_times:  # the gathered track holds the other ranks' slices too
            r2 = world - 1
            if ragged:
                a2, b2 = bounds[r2]
                iv2 = b2 - 1
                L2 = int(lens_all[iv2])
                c2 = int((lens_all[:iv2] + pad2 + 1).sum())
                s2 = int((lens_all[:iv2] + pad2 + 7).sum())
                o2 = int(lens_all[:iv2].sum())
            else:
                L2, g2 = L, r2 * n_iv + n_iv - 1
                c2, s2, o2 = g2 * l, g2 * (l + 6), g2 * L
            cp, cm = oracle.synth_counts(1, c2, L2 + pad2 + 1, 0), oracle.synth_counts(1, c2, L2 + pad2 + 1, 1)
            sq = oracle.synth_bases(1, s2, L2 + pad2 + 7)
            p2 = oracle.detect_batch(cp, cm, sq, 1, L2, HW, SHW, CLIP, table, DM.mu_params, DM.r_params, scales)[2]
            g2p = d_gather.download(np.float64, L2, o2 * 8)
            parity["gathered_last_rank_p_max_rel_err"] = float(
                np.nanmax(np.abs(g2p - p2) / np.maximum(np.abs(p2), 1e-300)))

    # ---- robustness of memo mode: tiles the first pass handed on, and (N=1) the same job with
    #      hotspot bursts in 2 % of the intervals, timed beside the headline
    robust = None
    if rank == 0 and args.nb_mode == "memo":
        tiles, redone, miss = ctx.scan_stats()
        robust = dict(hotspot_per_mille=args.hotspots, tiles=tiles, tiles_redone=redone,
                      largest_pair_outside_first_table=list(miss))
    heavy = sparse = None
    if world == 1 and not ragged and not args.no_heavy and not args.hotspots and args.nb_mode == "memo":
        pm = 20
        sc.synth_hotspots_dev(1, n_iv, L, p_cp, p_cm, pm)
        kh = max(3, args.steps // 4)
        # cold: the second-level (exp, obs) table emptied before every call -- every call computes it
        # between its passes and sends the hotspot tiles through the general kernel
        cold_tables[0] = True
        dtc = measure(kh, 1)[0]
        tiles_c, redone_c, miss_c = ctx.scan_stats()
        cold_tables[0] = False
        # kept (how the library runs): the table filled by the warm-up call stays, the first pass
        # reads it; what is still redone are tiles with pairs beyond its 4096 x 4096 entries
        dth = measure(kh, 1)[0]
        tiles, redone, miss = ctx.scan_stats()
        head = total_all * args.steps / dt
        heavy = dict(workload=cfg["name"] + "+hotspots_%dpermille" % pm, value=total * kh / dth, unit="bases/s",
                     ms_per_step=dth / kh * 1e3, steps=kh, tiles=tiles, tiles_redone=redone,
                     redone_fraction=redone / max(tiles, 1), largest_pair_outside_kept_tables=list(miss),
                     ratio_to_headline=(total * kh / dth) / head,
                     cold=dict(value=total * kh / dtc, ms_per_step=dtc / kh * 1e3, tiles_redone=redone_c,
                               redone_fraction=redone_c / max(tiles_c, 1),
                               largest_pair_outside_first_table=list(miss_c),
                               ratio_to_headline=(total * kh / dtc) / head),
                     note="observed counts up to ~1000 in the hotspots.  The second-level (exp, obs) table "
                          "(a function of the dispersion model alone) is kept by the context across calls "
                          "and grows to the largest pair met: `cold` is a call that meets the hotspots with "
                          "an empty table (computes it between the passes, sized on the device by the largest "
                          "pair the first pass missed, and redoes those tiles), the main figures are the calls "
                          "after it")

        # ---- the same job on SPARSE counts (Poisson 0.05 per base and strand: real data away from
        #      hotspots), which exercise the first pass differently: single cuts make runs of equal window
        #      sums.  A host-made block of counts is tiled over the resident arrays.
        blk_iv = min(n_iv, 20000)
        rs_sp = np.random.RandomState(7)
        blocks = [rs_sp.poisson(0.05, blk_iv * l).astype(np.float64) for _ in range(2)]
        for ptr, blk in ((p_cp, blocks[0]), (p_cm, blocks[1])):
            for a0 in range(0, n_iv, blk_iv):
                nb_ = min(blk_iv, n_iv - a0) * l
                _lib.check(ctx.L.fpt_memcpy_h2d(ctx.h, ptr + a0 * l * 8, blk.ctypes.data, nb_ * 8))
        dts = measure(kh, 1)[0]
        tiles_s, redone_s, _ = ctx.scan_stats()
        sparse = dict(workload=cfg["name"] + "+poisson0.05_counts", value=total * kh / dts, unit="bases/s",
                      ms_per_step=dts / kh * 1e3, steps=kh, tiles=tiles_s, tiles_redone=redone_s,
                      redone_fraction=redone_s / max(tiles_s, 1), ratio_to_headline=(total * kh / dts) / head)

    # ---- config 5, second half: the posterior footprint caller (cli/post.py:98-124) over D datasets
    #      of this rank's interval set, as ONE launch of fpt_posterior_dev.  Each dataset's tracks
    #      (exp, obs, empirical FDR) come from a scan + FDR pass over its own synthetic cut counts
    #      (seed 1 + d), made once outside the timed region; the launch is then timed K times.
    post = None
    D = cfg.get("posterior_datasets", 0)
    if D and fdr_times and not args.no_posterior:
        from footprint_tools_amd.stats import posterior as post_mod
        d_tracks = DeviceArray(ctx, 4 * D * total * 8)   # obs, exp, fdr, w: (D, total) each
        d_post = DeviceArray(ctx, D * total * 8)
        d_st = DeviceArray(ctx, max(n_iv, 1) * 4).zero()
        row = total * 8
        p_obs, p_exp, p_fdr, p_w = (d_tracks.ptr + k * D * row for k in range(4))
        ones = np.ones(total)
        for d_ in range(D):
            _lib.check(ctx.L.fpt_synth_dev(ctx.h, 1 + d_, pos0_counts, n_counts, p_cp, p_cm, pos0_seq, n_seq, p_sq))
            sc.scan_dev(n_iv, p_cp, p_cm, p_sq, exp_out=p_exp + d_ * row, obs_out=p_obs + d_ * row, pval_out=p_p,
                        winp_out=p_out + 2 * t8, interval_off_dev=d_off.ptr, interval_off_host=off,
                        dm_ids_dev=d_dm.ptr if d_dm else None)
            sc.fdr_dev(n_iv, p_exp + d_ * row, p_out + 2 * t8, p_fdr + d_ * row, times=fdr_times, seed=1 + d_,
                       half_win_width=scales[0], interval_off_dev=d_off.ptr, base_index0=bases_before,
                       dm_ids_dev=d_dm.ptr if d_dm else None, obs=p_obs + d_ * row, interval_off_host=off)
            _lib.check(ctx.L.fpt_memcpy_h2d(ctx.h, p_w + d_ * row, ones.ctypes.data, ones.nbytes))
        ctx.synchronize()
        # dataset d is scored with model d % n_models (the reference has one model per dataset)
        ds_models = [models[d_ % n_models] for d_ in range(D)] if n_models > 1 else [DM] * D
        packed = post_mod.pack_models(ds_models)  # handed over with every call (any number of datasets)
        betas = np.array([[2.0 + 0.5 * d_, 8.0 - 0.5 * d_] for d_ in range(D)])

        def post_step():
            post_mod.posterior_dev(ctx, n_iv, total, D, 0, betas, p_obs, p_exp, p_fdr, p_w, d_post.ptr,
                                   interval_off_dev=d_off.ptr, max_interval_len=int(lens.max()), fdr_cutoff=0.05,
                                   half_win_width=3, status_out=d_st.ptr, models=packed)
        post_step()
        ctx.synchronize()
        kp = max(3, args.steps // 2)
        t0 = time.perf_counter()
        pm = [ctx.mark()]
        for _ in range(kp):
            post_step()
            pm.append(ctx.mark())
        ctx.synchronize()
        dtp = (time.perf_counter() - t0) / kp
        post_ms = [ctx.mark_elapsed(pm[i], pm[i + 1]) for i in range(kp)]  # HIP events on the launch stream
        ctx.marks_clear()
        if comm is not None:
            dtp = comm.max_over_ranks(dtp)
        post = dict(n_datasets=D, launches=kp, ms_per_launch=dtp * 1e3, ms_per_launch_hip_events=float(np.mean(post_ms)),
                    value=total_all / dtp, unit="bases/s",
                    dataset_bases_per_s=total_all * D / dtp,
                    bound="fp64 vector ALU: 2 lgam + log + log1p + two piecewise fits (the occupied form; the unoccupied one "
                          "and lgam(k + 1) from tables made inside the call) + exp + log1p per dataset-base, 48 B of HBM "
                          "traffic per dataset-base (DESIGN.md, posterior kernel)",
                    hbm_GBps=total * D * 48.0 / dtp / 1e9,
                    # an operation count, not a counter: lgam ~45 flops x 2, log / log1p / exp / log1p ~25 each, the
                    # fits and p ~15, 2 x 7 window additions, the priors ~30 per dataset -- against the 78.6 TFLOP/s
                    # of the vector fp64 pipes (the kernel is bound by their instruction issue, divergent branches
                    # of lgam included, not by flops)
                    flops_per_dataset_base=270,
                    fp64_TFLOPs=total_all * D * 270.0 / dtp / 1e12, frac_of_fp64_vector_peak=total_all * D * 270.0 / dtp / 78.6e12)
        if rank == 0:  # one interval against the checker's restatement of cli/post.py:109-122
            from oracle import oracle
            iv = n_iv - 1
            a, b = int(off[iv]), int(off[iv + 1])
            host = [np.stack([d_tracks.download(np.float64, b - a, (k * D + d_) * row + a * 8) for d_ in range(D)])
                    for k in range(4)]
            want = oracle.posterior_stats(host[0], host[1], host[2], host[3], betas,
                                          [(m.mu_params, m.r_params) for m in ds_models], cutoff=0.05, hw=3)[0]
            got = d_post.download(np.float64, (b - a) * D, a * D * 8).reshape(b - a, D)
            post["parity_max_abs_err"] = float(np.nanmax(np.abs(got - want)))
            post["parity_ok"] = bool(np.allclose(got, want, rtol=1e-6, atol=1e-9, equal_nan=True))
            post["zero_division_flags"] = int(d_st.download(np.int32, n_iv).any())

    # ---- the box beside the kernel: the scan's loads and stores without its arithmetic
    #      (fpt_stream_pattern_dev: same layout, same workgroup per interval, 3 + S tracks written), timed
    #      here, after everything that needs the output buffers
    box = None
    if world == 1 and not args.no_box_stream:
        import ctypes
        ms = ctypes.c_float(0.0)
        _lib.check(ctx.L.fpt_stream_pattern_dev(ctx.h, n_iv, 0 if ragged else L, d_off.ptr if ragged else None,
                                                int(lens.max()) if ragged else L, HW + SHW, 3 + S, p_cp, p_cm, p_sq,
                                                p_out, total, 5, ctypes.byref(ms)))
        box = float(ms.value)

    if rank == 0:
        rd, wr = algorithmic_bytes_per_base(L if not ragged else total / n_iv, S)
        k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else None
        roof = None
        if k_ms:
            achieved = total * (rd + wr) / (k_ms * 1e-3) / 1e9
            # HBM bytes per launch measured with rocprofv3 --pmc (separate run, committed under
            # profiles/), expressed like `achieved`: bytes per launch / this run's kernel time
            traffic = traffic_bytes = None
            traffic_source = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if live_traffic:
                traffic_bytes = live_traffic["bytes_per_launch"]
                traffic_source = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE over two child runs of this command "
                                  "(2 + 1 steps each) made by this invocation: %.4g GB read (FETCH_SIZE doubled: gfx950) + "
                                  "%.4g GB written per step, divided by this run's kernel_ms"
                                  % (live_traffic["read_bytes"] / 1e9, live_traffic["write_bytes"] / 1e9))
            elif os.path.exists(tf):
                rec = json.load(open(tf)).get(cfg["name"])
                if rec:
                    traffic_bytes = rec["bytes_per_launch"]
                    traffic_source = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE of this "
                                      "workload in a separate run (gfx950 corrections applied), divided by "
                                      "this run's kernel_ms")
            if traffic_bytes:
                traffic = traffic_bytes / (k_ms * 1e-3) / 1e9
            achieved_read = total * rd / (k_ms * 1e-3) / 1e9
            roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                        # north_star's wording is the HBM-READ roofline: the same launch priced on
                        # its algorithmic read bytes only (frac above counts reads + writes)
                        achieved_read=achieved_read, frac_read=achieved_read / HBM_PEAK_GBS,
                        kernel=("k_scan_lean<NT, BPL> (first pass of the step; tiles outside its case are redone by "
                                "k_scan_fused<NT,5,50,table=L2,full>)"
                                if args.nb_mode == "memo" else "k_scan_fused<NT,HW,SHW,table=L2,full>"),
                        # the same loads and stores with no arithmetic between them, on this box, in this
                        # invocation (fpt_stream_pattern_dev): what the access pattern reaches here
                        box_stream_ms=box, box_stream_GBps=(total * (rd + wr) / (box * 1e-3) / 1e9 if box else None),
                        frac_of_box=(box / k_ms if box else None),
                        # GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration of the first-pass kernel in a child run
                        # under rocprofv3 --pmc (traffic_probe)
                        shader_clock_GHz=(live_traffic or {}).get("shader_clock_GHz"),
                        # how busy the vector and the LDS pipes were under the first-pass kernel (the same child run:
                        # SQ_ACTIVE_INST_VALU x 4 / (1,024 SIMDs x cycles), SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)): the
                        # HBM fraction above cannot pass what vector issue leaves
                        valu_busy=((live_traffic or {}).get("issue") or {}).get("valu_busy"),
                        lds_busy=((live_traffic or {}).get("issue") or {}).get("lds_busy"),
                        raw_counters=(live_traffic or {}).get("raw_counters"),
                        kernel_ms=k_ms, launch_sequence_ms=float(np.mean(seq_ms)),
                        algorithmic_bytes_per_launch=total * (rd + wr),
                        traffic_bytes_per_launch=traffic_bytes,
                        # measured HBM bytes over the algorithmic ones: ~1.0 says nothing is read or written twice
                        traffic_over_algorithmic=(traffic_bytes / (total * (rd + wr)) if traffic_bytes else None),
                        # HBM bytes per launch from the PMC counters -- read by this invocation in two
                        # child runs under rocprofv3 (traffic_probe), else the same measurement made
                        # earlier (profiles/traffic.json) -- over this run's kernel time
                        traffic_source=traffic_source,
                        algorithmic_bytes_per_base=dict(read=rd, write=wr))
        fdr_block = None
        if fdr_times and fdr_ms:
            # config 5: the step is the scan (k_ms of it, above) and the empirical-FDR pass; the dominant kernels are the
            # pass's null draws -- bound by vector / LDS instruction issue, priced on the SQ counters of a child run
            f_ms = float(np.mean(fdr_ms))
            draws = (issue or {}).get("draws")
            prof_total = sum(g["ms_per_call_under_profiler"] for g in issue.values()) if issue else None
            draws_ms = f_ms * draws["ms_per_call_under_profiler"] / prof_total if draws and prof_total else None
            clock = (draws or {}).get("shader_clock_GHz") or 2.4
            peak = 1024.0 * clock / 4.0  # G wave64 vector instructions per second: 1,024 SIMDs, 4 cycles each
            fdr_block = dict(ms_per_call=f_ms, draws_per_base=fdr_times, draws_per_s=total * fdr_times / (f_ms * 1e-3),
                             timing="HIP events on the launch stream around every fpt_fdr_dev of the timed region (fpt_mark)",
                             share_of_step=f_ms / (dt / args.steps * 1e3), kernel_groups=issue,
                             draws_ms=draws_ms)
            scan_roof = roof
            # (three units near their limits at once, none alone: the vector pipes, LDS, and the texture addresser that
            # the sampler's two table gathers per draw go through -- `frac` stays the vector pipes' as the review of
            # round 4 defined it, the other two ride beside it)
            # A work-based fraction (round 6; rounds 4 - 5 printed VALUBusy here, which any kernel that burns vector
            # instructions scores high on): draws per second over the rate at which the vector pipes could issue a
            # FLOOR of instructions per draw -- Philox4x32-10 15 (sixty per block of four words), the alias draw 4 (slot,
            # address, compare, select), the 7-term window sum and its negation 7, the guide slice 3 (fused multiply-add,
            # clamp, convert), one threshold probe 2, the histogram add 1: 32 per draw and lane, against the 86 the
            # kernel issues (`valu_instructions_per_draw`).  The gathers and LDS accesses of the same floor (2 and ~6 per
            # draw) fit beside them; VALUBusy stays in the block as `valu_busy`.
            floor_instr = 32.0
            floor_draws_per_s = peak * 1e9 * 64.0 / floor_instr
            draws_rate = (total * fdr_times / (draws_ms * 1e-3)) if draws_ms else None
            roof = dict(bound="valu+lds issue",
                        also_near_its_limit="the texture addresser (the sampler's two table gathers per draw): ta_busy",
                        kernel="the null draws of fpt_fdr_dev: k_fdr_null<NT,false,3,true,3> (one workgroup per interval of up to "
                               "256 bases) + k_fdr_slice<192> (slices of longer intervals); set-up k_fdr_null<...,1> before them",
                        achieved=(draws_rate / 1e9 if draws_rate else None),
                        peak=floor_draws_per_s / 1e9, unit="G null draws/s",
                        frac=(draws_rate / floor_draws_per_s if draws_rate else None), traffic=None,
                        frac_is="draws/s of the draw kernels over the vector-issue rate of a 32-instruction draw (Philox 15, alias "
                                "draw 4, window 7, guide slice 3, one probe 2, histogram 1) at 1,024 SIMDs x clock / 4",
                        floor_valu_instructions_per_draw=floor_instr,
                        valu_G_instructions_per_s=(draws["valu_instructions_per_call"] / (draws_ms * 1e-3) / 1e9
                                                   if draws and draws_ms else None),
                        valu_peak_G_instructions_per_s=peak,
                        # rocprofv3's VALUBusy of the draw kernels: SQ_ACTIVE_INST_VALU x 4 / (1,024 SIMDs x cycles)
                        valu_busy=(draws["valu_busy"] if draws else None),
                        lds_busy=(draws["lds_busy"] if draws else None),
                        ta_busy=(draws.get("ta_busy") if draws else None),
                        ta_busy_cycles_per_gather_instruction=(draws.get("ta_busy_cycles_per_vmem_read_instruction") if draws else None),
                        lds_bank_conflict_cycles_per_lds_instruction=(draws["lds_bank_conflict_cycles_per_lds_instruction"]
                                                                      if draws else None),
                        valu_instructions_per_draw=(draws["valu_instructions_per_call"] * 64.0 / (total * fdr_times)
                                                    if draws else None),
                        kernel_ms=draws_ms, fdr_pass_ms=f_ms, shader_clock_GHz=clock,
                        source=("one child run of this command under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
                                "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE, a second one with TA_TA_BUSY_sum "
                                "SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE (issue_probe); kernel_ms = "
                                "the HIP-event time of the pass x the draw kernels' share of its dispatch time in that run"
                                if issue else "no counters in this run (rocprofv3 missing, or the run is itself profiled)"),
                        scan=scan_roof)
        out = {
            "metric": "bases/sec per-nucleotide footprint stats",
            "value": total_all * args.steps / dt,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg["name"] + ("+hotspots_%dpermille" % args.hotspots if args.hotspots else ""),
                       "intervals_per_gpu": n_iv if not ragged else cfg["n_iv"],
                       "interval_bp": L if not ragged else "lognormal, mean %.0f, [50,2000]" % (total_all / lens_all.size),
                       "half_win_width": HW, "smoothing_half_win_width": SHW, "smoothing_clip": CLIP,
                       "stouffer_half_widths": list(scales), "bias_model": "vierstra_et_al.6mer",
                       "dispersion_model": "DM-SYNTH-A" if n_models == 1 else "%d per-interval variants of DM-SYNTH-A" % n_models,
                       "empirical_fdr_null_draws_per_base": fdr_times or None,
                       "nb_pvalue": ("exact (exp,obs)->(p,z) memo table, 256x256, rebuilt by the device "
                                     "incbet inside every step; second-level table / direct incbet outside it"
                                     if args.nb_mode == "memo" else "direct incbet per base"),
                       "sharding": ("one global interval list cut into contiguous ranges balanced by padded bases"
                                    if ragged else "rank r owns intervals [r*n, (r+1)*n) of the global job"),
                       "allgather_p_track": bool(do_gather),
                       "allgather": ("RCCL (librccl.so bound by libfpt_hip, no torch), %s, once after the %d steps, "
                                     "inside the timed region" % (("to rank 0: grouped ncclSend / ncclRecv" if to_root else
                                                                   "ragged shards: grouped ncclBroadcast" if ragged
                                                                   else "ncclAllGather"), args.steps))
                       if do_gather else None},
            "roofline": roof,
            # `value` above: K scans, then ONE assembly of the resident batch's track, inside the timed region
            # (north_star: "a single RCCL all-gather at the end").  Beside it, separately: the scans alone
            # (what weak-scaling efficiency of the path itself is computed from), and a job that assembles
            # EVERY step's track on the communicator's own stream while the next step is scanned
            "multi_gpu": (dict(assembly=args.assembly if do_gather else "none", steps=args.steps,
                               scan_s=dt_scan, allgather_s=dt_gather, allgather_bytes_per_rank=total_all * 8,
                               allgather_GBps_per_rank=(total_all * 8 / dt_gather / 1e9 if dt_gather > 0 else None),
                               assembled_bytes_received=dict(
                                   allgather_every_rank=(total_all - total) * 8, gather_root=(total_all - counts[0]) * 8),
                               scan_only=dict(value=total_all * args.steps / dt_scan, unit="bases/s",
                                              ms_per_step=dt_scan / args.steps * 1e3),
                               with_assembly=(dict(value=total_all * args.steps / dt_asm, unit="bases/s",
                                                   ms_per_step=dt_asm / args.steps * 1e3, overlapped=True,
                                                   note="every step's track assembled (%s) on the communicator's stream beside "
                                                        "the next step's scan, two track buffers in turn" % args.assembly)
                                              if dt_asm else None),
                               bases_per_rank=counts,
                               # how to read `value` against N x the one-GPU value: the K scans are
                               # communication-free, the ONE assembly at the end moves this rank's shard to every peer over
                               # that peer's own xGMI link (~50 GB/s one way assumed for a large transfer), so a perfectly
                               # scaling job still reads K t_scan / (K t_scan + shard_bytes / link) of linear in `value`;
                               # `scan_only` is the figure weak-scaling efficiency of the path itself is computed from
                               expected_value_vs_linear=(
                                   dt_scan / (dt_scan + max(counts) * 8 / 50e9) if (do_gather and world > 1) else 1.0),
                               expected_value_vs_linear_assumes=dict(link_GBps_one_way=50.0, shard_bytes=max(counts) * 8,
                                                                     assemblies_in_timed_region=1 if do_gather else 0),
                               value_over_scan_only=(dt_scan / dt if dt > 0 else None),
                               # ncclCommCount / ncclCommUserRank / ncclCommCuDevice of every rank's communicator and
                               # the PCI bus id of its device, gathered over the communicator (fpt_comm_info)
                               rccl_ranks=(min(j["rccl_count"] for j in job) if job else None), ranks=job,
                               distinct_devices=(len(set(j["pci_bus_id"] for j in job)) if job else None))
                          if (world > 1 or do_gather) else None),
            "cpu_baseline": base,
            "other_nb_mode": other,
            "memo_robustness": robust,
            "heavy_tailed": heavy,
            "sparse_counts": sparse,
            "posterior": post,
            "fdr": fdr_block,
            "parity": parity,
            "host_arrays": host_leg,
            "degraded": degraded + (["host_arrays leg: " + host_leg["error"]] if host_leg and "error" in host_leg else []),
        }
        if legs is not None:
            # every BASELINE configuration that fits one GPU, from THIS invocation: the headline (this process) and
            # configs 2, 4, 5 as child runs (config 4 / 5: one GPU's share, 437,500 ragged intervals)
            me = leg_summary(out)
            out["configs"] = {"2": legs.get("2"), "3": me, "4": legs.get("4"), "5": legs.get("5")}
        # the figures of every block once more, compact and LAST in the line: a reader that keeps only the tail of a
        # long line still sees every configuration, the host-arrays leg and the CPU figures
        def brief(leg):
            if not leg or "error" in leg:
                return leg
            r_ = leg.get("roofline") or {}
            b = dict(ms_per_step=round(leg["ms_per_step"], 4), value=float("%.4g" % leg["value"]),
                     frac=(round(r_["frac"], 4) if r_.get("frac") is not None else None), bound=r_.get("bound"),
                     kernel_ms=(round(r_["kernel_ms"], 4) if r_.get("kernel_ms") else None),
                     valu_busy=(round(r_["valu_busy"], 3) if r_.get("valu_busy") else None),
                     exp_bit_exact=(leg.get("parity") or {}).get("exp_bit_exact"),
                     p_max_rel_err=(leg.get("parity") or {}).get("p_max_rel_err"),
                     winp_max_rel_err=(leg.get("parity") or {}).get("winp_max_rel_err"))
            if (leg.get("parity") or {}).get("efdr_max_abs_err") is not None:
                b["efdr_max_abs_err"] = leg["parity"]["efdr_max_abs_err"]
            if leg.get("posterior"):
                b["posterior_dataset_bases_per_s"] = float("%.4g" % leg["posterior"]["dataset_bases_per_s"])
                b["posterior_parity_ok"] = leg["posterior"].get("parity_ok")
            return b
        summ = dict(configs=({k: brief(v) for k, v in out["configs"].items()} if "configs" in out else {args.config: brief(out)}))
        if host_leg and "error" not in host_leg:
            summ["host_arrays_bases_per_s"] = dict(pageable=float("%.4g" % host_leg["pageable"]["value"]),
                                                   pinned=float("%.4g" % host_leg["pinned"]["value"]),
                                                   pageable_new_output_arrays=float("%.4g" % host_leg["pageable"].get(
                                                       "new_output_arrays_value", 0.0)),
                                                   parity_ok=bool(host_leg["parity"]["exp_bit_exact"]
                                                                  and host_leg["parity"]["p_max_rel_err"] < 1e-6
                                                                  and host_leg["parity"]["pinned_equals_pageable_bitwise"]))
        if base:
            summ["cpu_bases_per_s"] = dict(port_allcores=float("%.4g" % base["value"]), port_1core=float("%.4g" % base["value_1core"]),
                                           reference_native_allcores=base.get("reference_native_allcores"),
                                           reference_native_1core=base.get("reference_native_1core"), cores=base["cores"])
        summ["degraded"] = len(out["degraded"])
        out["summary"] = summ
        if args.share_gpu and world > 1:  # a smoke test of the launcher path, not a measurement
            out["invalid"] = ("--share-gpu: %d ranks on ONE GPU (%s): not a scaling measurement" % (
                world, "collectives through the test suite's librccl stand-in" if comm is not None else
                "no communicator, ranks neither synchronised nor their times combined"))
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
