"""Timeline of the lean scan kernel's workgroups from the ablation build's trace
(FPT_LEAN_TRACE=<file>, libfpt_hip_ablate.so): per-phase durations of a workgroup, and the idle
gap between one workgroup's end and the start of the one that takes its place on the same CU."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64).reshape(-1, 8)
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 1024  # workgroup size -> workgroups resident per CU
slots = {128: 10, 192: 8, 256: 6, 384: 5, 512: 4, 768: 2, 1024: 2}.get(nt, 2)
hw = raw[:, 0] & 0xffffffff
xcc = (raw[:, 0] >> 32) & 0xf
cu = (xcc << 8) | ((hw >> 8) & 0xff)  # xcc, se, sh, cu
t = raw[:, 1:8].astype(np.float64) * 0.01  # microseconds
t0 = t[:, 0].min()
print("workgroups %d  CUs seen %d  kernel span %.1f us" % (len(raw), len(np.unique(cu)), t[:, 5:7].max() - t0))
names = ["load wait (start -> inputs staged)", "barrier 1", "phase B + barrier 2", "phases C, D", "phase E (first wave)"]
d = np.diff(t[:, :6], axis=1)
for i, n in enumerate(names):
    print("  %-38s mean %6.2f  median %6.2f  p90 %6.2f us" % (n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
end = np.maximum(t[:, 5], t[:, 6])
life = end - t[:, 0]
print("  %-38s mean %6.2f  median %6.2f  p90 %6.2f us" % ("life (start -> last store issued)", life.mean(), np.median(life), np.percentile(life, 90)))
gaps = []
busy = []
for c in np.unique(cu):
    m = np.where(cu == c)[0]
    o = m[np.argsort(t[m, 0])]
    ends = []
    for k in o:
        s = t[k, 0]
        if len(ends) >= slots:
            e = min(ends)
            ends.remove(e)
            gaps.append(s - e)
        ends.append(end[k])
    span = end[o].max() - t[o, 0].min()
    busy.append(life[o].sum() / (span * slots))
gaps = np.array(gaps)
print("  %-38s mean %6.2f  median %6.2f  p90 %6.2f us" % ("gap (end -> successor's start)", gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))
print("  slot occupancy (life / span / slots)   mean %.3f" % np.mean(busy))
per = np.bincount(cu.astype(np.int64))
per = per[per > 0]
print("  workgroups per CU: min %d max %d" % (per.min(), per.max()))
