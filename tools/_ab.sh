cd $GRAFT_REPO_ROOT
for i in 1 2; do
python3 tools/bench_ops.py gemm4 2>&1 | grep -v amdgpu.ids
echo "--- SVOL_WS_DEEP=1"
SVOL_WS_DEEP=1 python3 tools/bench_ops.py gemm4 2>&1 | grep -v amdgpu.ids
done
