set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r${RN:-5}
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r${RN:-5} -- python3 bench.py --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp $F $O/round${RN:-5}_kernel_stats.csv
python tools/prof_summary.py $F 16 > $O/round${RN:-5}_kernel_stats.txt
python tools/trace_timeline.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/round${RN:-5}_timeline.txt 2>&1 || true
python - <<PY > $O/round${RN:-5}_turnaround.txt
import csv,glob
f=glob.glob('$O/kt/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    r['s'],r['e']=int(r['Start_Timestamp']),int(r['End_Timestamp']); r['n']=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]
rows.sort(key=lambda r:r['s'])
ends=[r['e'] for r in rows if 'adamw_flat' in r['n']]
marks=sorted(set(ends))
step_ends=[marks[i] for i in range(len(marks)) if i+1==len(marks) or marks[i+1]-marks[i]>2000000]
t0,t1=step_ends[-2],step_ends[-1]
win=[r for r in rows if r['s']>=t0 and r['e']<=t1+1000]
# print every kernel between the last forward attention and the first sp backward, and around the step boundary
fa=[i for i,r in enumerate(win) if 'attn_fwd_bf16_fast' in r['n']]
sb=[i for i,r in enumerate(win) if 'attn_bwd_sp_bf16' in r['n'] and 'prep' not in r['n']]
print('--- forward -> backward turn (from the last video attention forward to the first single-pass backward) ---')
for r in win[fa[-1]:sb[0]+1]:
    print('%9.1f us +%7.1f  q%s  %s' % ((r['s']-win[fa[-1]]['s'])/1e3, (r['e']-r['s'])/1e3, r['Queue_Id'], r['n']))
print('--- step start (until the first video attention forward) ---')
for r in win[:fa[0]+1]:
    print('%9.1f us +%7.1f  q%s  %s' % ((r['s']-t0)/1e3, (r['e']-r['s'])/1e3, r['Queue_Id'], r['n']))
print('--- step end (after the last single-pass backward) ---')
for r in win[sb[-1]:]:
    print('%9.1f us +%7.1f  q%s  %s' % ((r['s']-win[sb[-1]]['s'])/1e3, (r['e']-r['s'])/1e3, r['Queue_Id'], r['n']))
PY
rm -rf $O/kt
head -50 $O/round${RN:-5}_kernel_stats.txt
cat $O/round${RN:-5}_timeline.txt | head -40
