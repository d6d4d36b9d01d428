cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_f; mkdir -p $O
run() { # label cfg
  python3 bench.py --config $2 --no-heavy --no-cpu-baseline --no-traffic-probe --no-other-mode --no-config-legs --no-issue-probe --no-posterior --no-box-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 cfg$2 kernel_ms', round(d['roofline'].get('scan', d['roofline'])['kernel_ms'],4), 'step_ms', round(d['ms_per_step'],4), d.get('parity'))"
}
for rep in 1 2; do
  for cfg in 4 2 3; do
    for pf in 0 0.5 1 2; do
      FPT_LEAN_PREFETCH=$pf run pf$pf $cfg >> $O/prefetch.log 2>&1
    done
  done
done
cat $O/prefetch.log
bash tools/prof_post.sh r05a > $O/post.log 2>&1
tail -32 $O/post.log
