export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_chain
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_chain/p1 -o p1 -- python3 tools/lab_mlp_chain.py > gpurun_out/pmc_chain/p1.log 2>&1
echo p1 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_chain/p2 -o p2 -- python3 tools/lab_mlp_chain.py > gpurun_out/pmc_chain/p2.log 2>&1
echo p2 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --output-format csv -d gpurun_out/pmc_chain/p3 -o p3 -- python3 tools/lab_mlp_chain.py > gpurun_out/pmc_chain/p3.log 2>&1
echo p3 rc=$?
find gpurun_out/pmc_chain -name "*counter_collection.csv" | sort | while read f; do echo "== $f"; python tools/pmc_summary.py $f mlp_chain; done > gpurun_out/pmc_chain/summary.txt
cat gpurun_out/pmc_chain/summary.txt
