# kernel statistics of the attention backward launch group with (on) / without (off) SVOL_DQ_PREZERO=1:  bash tools/kt_prezero.sh  -> gpurun_out/kt_{on,off}/
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for V in on off; do
  O=$R/gpurun_out/kt_$V; rm -rf $O; mkdir -p $O
  if [ $V = off ]; then unset SVOL_DQ_PREZERO; else export SVOL_DQ_PREZERO=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python3 bench.py --steps 6 --warmup 3 --blocks 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
  F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  grep -E "sp_prep|sp_zero|attn_bwd_sp_bf16|dq_round" $F > $O/stats.txt || true
  T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  python3 - "$T" > $O/around.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'],r['e']=int(r['Start_Timestamp']),int(r['End_Timestamp']); r['n']=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:50]
rows.sort(key=lambda r:r['s'])
idx=[i for i,r in enumerate(rows) if r['n'].startswith('attn_bwd_sp_bf16')]
i=idx[-3]
t0=rows[i]['s']
for r in rows[i-8:i+12]:
    print('%9.1f us +%7.1f  q%s  %s' % ((r['s']-t0)/1e3,(r['e']-r['s'])/1e3,r['Queue_Id'],r['n']))
PY
  rm -rf $O/kt
  cat $O/stats.txt; cat $O/around.txt
done
