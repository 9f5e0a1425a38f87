set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traffic_r${RN:-5}
rm -rf $O; mkdir -p $O
cd $R
HEAD=$1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline > $O/w.log 2>&1
python tools/hbm_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/round${RN:-5}_hbm_traffic $HEAD > /dev/null
python tools/fwd_traffic.py $(find $O/fetch -name "*kernel_trace.csv" | head -1) $O/round${RN:-5}_hbm_traffic.json 343 "cfg2 forward (bf16, B=8)" > $O/round${RN:-5}_cfg2_fwd_traffic.txt 2>&1 || true
rm -rf $O/fetch $O/write
head -30 $O/round${RN:-5}_hbm_traffic.txt
cat $O/round${RN:-5}_cfg2_fwd_traffic.txt | head -30
# kernel stats of the same build
bash tools/prof_step.sh > $O/prof.log 2>&1 || true
