cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_n; mkdir -p $O
show() { python3 -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1]); h=d['host_arrays']
print('$2', {k:(round(h[k]['value']/1e9,3), round(h[k]['ms_per_call'],1), h[k]['calling_thread_ms']) for k in ('pageable','pinned')})"; }
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic-probe --no-config-legs --no-other-mode --no-heavy --no-box-stream > $O/a.json 2>/dev/null; show $O/a.json "no-cpu-baseline,no-probes,no-legs:"
python3 bench.py --steps 3 --warmup 1 --no-traffic-probe --no-config-legs --no-other-mode --no-heavy --no-box-stream > $O/b.json 2>/dev/null; show $O/b.json "with cpu baseline:"
python3 bench.py --steps 3 --warmup 1 --intervals 100000 --no-cpu-baseline --no-traffic-probe --no-config-legs --no-other-mode --no-heavy --no-box-stream > $O/c.json 2>/dev/null; show $O/c.json "small resident batch (100k intervals):"
python3 tools/bench_host_arrays.py 0 | grep chunk
