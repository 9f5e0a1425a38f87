# Re-measure every profiles/round${RN}_* file at ONE git head (the r4 traffic file was two kernel commits old).
#   usage (GPU box): RN=5 bash tools/refresh_profiles.sh <git-head>      -> gpurun_out/refresh_r${RN}/  (copy into profiles/)
# rocprofv3 rule of the pool: the program itself after "--" (python3 bench.py ...), counters and --stats in separate runs.
export TMPDIR=/tmp
RN=${RN:-5}
HEAD=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh_r$RN
rm -rf $O; mkdir -p $O
cd $R
echo "== default bench"
python bench.py --steps 20 --warmup 5 > $O/round${RN}_bench.json 2> $O/bench.err
echo "== traffic + kernel stats (cfg2)"
RN=$RN bash tools/traffic_step.sh $HEAD > $O/traffic.log 2>&1
cp gpurun_out/traffic_r$RN/round${RN}_* $O/ 2>/dev/null
cp gpurun_out/prof_r$RN/round${RN}_* $O/ 2>/dev/null
echo "== block trace (no profiler)"
SVOL_BLOCK_TRACE=1 python tools/block_trace.py 10 > $O/round${RN}_block_trace.txt 2>&1
SVOL_BLOCK_TRACE=2 python tools/block_trace.py 6 > $O/round${RN}_block_timeline.txt 2>&1
echo "== cfg5 fp16: traffic, kernel stats, roofline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch5 -o f -- python3 bench.py --workload cfg5 --dtype fp16 --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $O/f5.json 2> $O/f5.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write5 -o w -- python3 bench.py --workload cfg5 --dtype fp16 --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $O/w5.json 2> $O/w5.err
python tools/hbm_traffic.py $(find $O/fetch5 -name "*counter_collection.csv") $(find $O/write5 -name "*counter_collection.csv") $O/round${RN}_cfg5_fp16_hbm_traffic $HEAD > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -o r5 -- python3 bench.py --workload cfg5 --dtype fp16 --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline > $O/kt5_bench.json 2> $O/kt5_bench.err
cp $(find $O/kt5 -name "*kernel_stats.csv" | head -1) $O/round${RN}_cfg5_fp16_kernel_stats.csv
python tools/fwd_traffic.py $(find $O/kt5 -name "*kernel_trace.csv" | head -1) $O/round${RN}_cfg5_fp16_hbm_traffic.json 224 "cfg5 (B=1, T=128, P=256, L=32768, fp16)" > $O/round${RN}_cfg5_fp16_roofline.txt 2>&1 || true
rm -rf $O/fetch5 $O/write5 $O/kt5
echo "== the other bench modes"
python bench.py --workload cfg5 --dtype fp16 --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_cfg5_fp16.json 2>> $O/bench.err
python bench.py --workload cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_cfg5_bf16.json 2>> $O/bench.err
python bench.py --dtype fp16 --steps 20 --warmup 5 --no-cpu-baseline > $O/round${RN}_bench_fp16.json 2>> $O/bench.err
python bench.py --dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline > $O/round${RN}_bench_fp32.json 2>> $O/bench.err
python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_cfg4.json 2>> $O/bench.err
python bench.py --workload encdec --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_encdec.json 2>> $O/bench.err
python bench.py --workload resnet --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_resnet.json 2>> $O/bench.err
python bench.py --workload resnet --train-backbone --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_resnet_trained.json 2>> $O/bench.err
python bench.py --workload encdec --dropout 0.1 --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_encdec_dropout.json 2>> $O/bench.err
SVOL_DETERMINISTIC=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/round${RN}_bench_deterministic.json 2>> $O/bench.err
echo "== trained-backbone step: kernel statistics"
bash tools/prof_resnet_train.sh > $O/prof_resnet_train.log 2>&1
cp gpurun_out/prof_rn/kernel_stats.txt $O/round${RN}_resnet_train_kernel_stats.txt
ls $O
