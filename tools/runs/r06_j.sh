cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_j; mkdir -p $O
F="--no-cpu-baseline --no-traffic-probe --no-other-mode --no-heavy --no-config-legs --no-issue-probe --no-box-stream --no-posterior --no-host-arrays"
rocprofv3 --kernel-trace --output-format csv -d $O/trace4 -- python3 bench.py --config 4 --steps 10 --warmup 50 $F > $O/trace4.log 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/r06_j/trace4/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('Workgroup_Size_X', r.get('Workgroup_Size',''))))
rows.sort()
# last 3 steps: find k_nb_memo starts
idx=[i for i,r in enumerate(rows) if 'k_nb_memo' in r[2] and 'memo2' not in r[2]]
for a,b in zip(idx[-4:-1], idx[-3:]):
    t0=rows[a][0]; prev=None
    print('--- step')
    for r in rows[a:b]:
        gap = (r[0]-prev)/1e3 if prev else 0.0
        print('%8.1f us  dur %7.1f us  gap %6.1f us  %s grid %s wg %s' % ((r[0]-t0)/1e3, (r[1]-r[0])/1e3, gap, r[2], r[3], r[4]))
        prev=r[1]
    print('step span %.1f us' % ((rows[b-1][1]-t0)/1e3))
PY
