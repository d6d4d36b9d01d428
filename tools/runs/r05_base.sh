cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_base
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05_base/pytest.log
for c in 3 2 4 5; do
  python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r05_base/bench_cfg$c.json 2> gpurun_out/r05_base/bench_cfg$c.err
done
tail -c 600 gpurun_out/r05_base/pytest.log
for c in 3 2 4 5; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r05_base/bench_cfg$c.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$c', d['ms_per_step'], d['value'], r['frac'], r['kernel_ms'], r.get('frac_of_box'), r.get('traffic'))
"; done
