# round 5: SQ counters of the hand-placed forward (attn_fwd_bf16_fast2) and round 2's (attn_fwd_bf16_fast, SVOL_ATTN_FWD_V1=1).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_fwd5
for v in new old; do
  if [ $v = old ]; then export SVOL_ATTN_FWD_V1=1; else unset SVOL_ATTN_FWD_V1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_fwd5/${v}1 -o p -- python3 tools/bench_ops.py attn > gpurun_out/pmc_fwd5/${v}1.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_fwd5/${v}2 -o p -- python3 tools/bench_ops.py attn > gpurun_out/pmc_fwd5/${v}2.log 2>&1
done
find gpurun_out/pmc_fwd5 -name "*counter_collection.csv" | sort | while read f; do echo "== $f"; python tools/pmc_summary.py $f attn_fwd_bf16_fast; done > gpurun_out/pmc_fwd5/summary.txt
cat gpurun_out/pmc_fwd5/summary.txt
