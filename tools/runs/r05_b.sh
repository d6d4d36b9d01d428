cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_b; mkdir -p $O
python3 -m pytest tests/test_ab_two_ranks.py -m gpu -x -q -k "many_ranks" 2>&1 | tail -60 > $O/pytest_ranks.log
python3 -m pytest tests -m gpu -x -q --deselect tests/test_ab_two_ranks.py::test_many_ranks_one_gpu_collectives 2>&1 | tail -30 > $O/pytest.log
python3 bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python3 tools/bench_posterior.py > $O/posterior.log 2>&1
python3 tools/bench_fdr_ragged.py 100000 4,52,100 > $O/fdr_ragged.log 2>&1
export FPT_LIB_PATH=$PWD/footprint_tools_amd/libfpt_hip_ablate.so
for bits in 0 512 1024 2048 4096 5120 7680; do
  echo -n "slices=0 ablate=$bits: " >> $O/ablate.log; FPT_FDR_SLICES=0 FPT_ABLATE=$bits python3 tools/bench_fdr_ragged.py 100000 100 2>&1 | tail -1 >> $O/ablate.log
done
unset FPT_LIB_PATH
cat $O/pytest_ranks.log | tail -40; cat $O/pytest.log | tail -15; cat $O/posterior.log | tail -3; cat $O/fdr_ragged.log; cat $O/ablate.log
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_b/bench_cfg5.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['parity'])
r=d['roofline']; print({k:r[k] for k in ('frac','lds_busy','valu_instructions_per_draw','kernel_ms','fdr_pass_ms','lds_bank_conflict_cycles_per_lds_instruction')})
p=d['posterior']; print({k:p[k] for k in ('ms_per_launch','ms_per_launch_hip_events','dataset_bases_per_s','parity_max_abs_err','parity_ok')})
for g,v in d['fdr']['kernel_groups'].items(): print(g, {k:(round(x,4) if isinstance(x,float) else x) for k,x in v.items()})
PY
