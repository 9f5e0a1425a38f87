# round 5: SQ counters of the K = 256 weight-stationary GEMMs of the video half's MLP (fc1 + gelu', dmul) — two passes.
# usage (GPU box): bash tools/pmc_ws5.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_ws5
python3 tools/bench_ops.py gemm4 > gpurun_out/pmc_ws5/times.txt 2>&1
cat gpurun_out/pmc_ws5/times.txt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_ws5/p1 -o p -- python3 tools/bench_ops.py gemm4 > gpurun_out/pmc_ws5/p1.log 2>&1
echo p1 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_ws5/p2 -o p -- python3 tools/bench_ops.py gemm4 > gpurun_out/pmc_ws5/p2.log 2>&1
echo p2 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_EXP_GDS --output-format csv -d gpurun_out/pmc_ws5/p3 -o p -- python3 tools/bench_ops.py gemm4 > gpurun_out/pmc_ws5/p3.log 2>&1
echo p3 rc=$?
find gpurun_out/pmc_ws5 -name "*counter_collection.csv" | sort | while read f; do echo "== $f"; python tools/pmc_summary.py $f gemm_ws; done > gpurun_out/pmc_ws5/summary.txt
cat gpurun_out/pmc_ws5/summary.txt
