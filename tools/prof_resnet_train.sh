# rocprofv3 kernel statistics of the trained-backbone step (bench.py --workload resnet --train-backbone); usage (GPU box): bash tools/prof_resnet_train.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_rn
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o e -- python3 bench.py --workload resnet --train-backbone --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench.err
F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $F 4 > $O/kernel_stats.txt
head -30 $O/kernel_stats.txt
