export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/bench_ops.py gate ln 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gate_kt -o g -- python3 tools/bench_ops.py gate ln > /dev/null 2>&1
python tools/prof_summary.py $(find gpurun_out/gate_kt -name "*kernel_stats.csv" | head -1) 1 | head -16
