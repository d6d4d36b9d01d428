"""The host-arrays leg of bench.py alone (numpy in / numpy out through FootprintScanner.scan, config 2's size,
PCIe included), for several chunk sizes of the pipeline (FPT_BENCH_HOST_CHUNK).  Diagnostic."""
import json
import os
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402

table, DM = bench.load_models()
from oracle import oracle  # noqa: E402
oracle.lib()
from footprint_tools_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
for chunk in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]:
    os.environ["FPT_BENCH_HOST_CHUNK"] = str(chunk)
    d = bench.host_arrays_leg(ctx, table, DM)
    for k in ("pageable", "pinned"):
        v = d[k]
        print("chunk %9d %-8s %.3g bases/s  call %.1f ms  first %.1f ms  chunks %d  link %.1f + %.1f GB/s  thread %s" % (
            chunk, k, v["value"], v["ms_per_call"], v["first_call_ms"], v["chunks"], v["link_GBps_h2d"], v["link_GBps_d2h"],
            json.dumps({a: round(b, 1) for a, b in v["calling_thread_ms"].items()})) +
              ("  new output arrays %.1f ms" % v["new_output_arrays_ms"] if "new_output_arrays_ms" in v else ""))
    print("   parity", d["parity"])
