export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_encdec
rm -rf $O; mkdir -p $O
python3 bench.py --workload encdec --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_p0.json 2> $O/err0.txt
python3 bench.py --workload encdec --dropout 0.1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_p1.json 2> $O/err1.txt
cat $O/bench_p0.json $O/bench_p1.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o e -- python3 bench.py --workload encdec --dropout 0.1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench.err
F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $F 6 > $O/kernel_stats.txt
head -40 $O/kernel_stats.txt
