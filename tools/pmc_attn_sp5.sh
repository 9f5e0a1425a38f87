# round 5: SQ counters of the re-placed single-pass backward (attn_bwd_sp_bf16), round 4's placement (attn_bwd_sp_bf16_v9) and the
# timing-only ablations of the new body, two passes.  usage (GPU box): bash tools/pmc_attn_sp5.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_sp5
export SP_ABLATIONS=1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sp5/p1 -o p1 -- tools/micro/attn_lab_sp5 2 > gpurun_out/pmc_sp5/p1.log 2>&1
echo p1 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sp5/p2 -o p2 -- tools/micro/attn_lab_sp5 2 > gpurun_out/pmc_sp5/p2.log 2>&1
echo p2 rc=$?
find gpurun_out/pmc_sp5 -name "*counter_collection.csv" | sort | while read f; do echo "== $f"; python tools/pmc_summary.py $f attn_bwd_sp; done > gpurun_out/pmc_sp5/summary.txt
cat gpurun_out/pmc_sp5/summary.txt
