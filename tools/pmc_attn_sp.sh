export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_sp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sp/p1 -o p1 -- tools/micro/attn_lab_sp 2 > gpurun_out/pmc_sp/p1.log 2>&1
echo p1 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sp/p2 -o p2 -- tools/micro/attn_lab_sp 2 > gpurun_out/pmc_sp/p2.log 2>&1
echo p2 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_sp/p3 -o p3 -- tools/micro/attn_lab_sp 2 > gpurun_out/pmc_sp/p3.log 2>&1
echo p3 rc=$?
find gpurun_out/pmc_sp -name "*counter_collection.csv" | sort | while read f; do echo "== $f"; python tools/pmc_summary.py $f attn_bwd; done > gpurun_out/pmc_sp/summary.txt
cat gpurun_out/pmc_sp/summary.txt
