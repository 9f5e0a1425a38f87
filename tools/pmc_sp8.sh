# round 6: SQ counters of the EMULATED two-waves-per-SIMD attention backward (tools/micro/gen_mfma_fillers.py: sp8_phase / sp8_fine = one
# step of one wave of the 8-wave x 64-key design at two waves per SIMD) beside the emulation of the CURRENT design (step4, one wave per
# SIMD; at two waves per SIMD it is not buildable: 451 registers) — the counter evidence VERDICT r5 item 2 asks for before (not) building it.
# usage (GPU box): bash tools/pmc_sp8.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc_sp8
python3 tools/micro/gen_mfma_fillers.py > tools/micro/mfma_fillers.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_fillers tools/micro/mfma_fillers.hip || exit 1
for v in step4 sp8_phase sp8_fine; do
  tools/micro/mfma_fillers $v | tail -1
done > gpurun_out/pmc_sp8/timing.txt
cat gpurun_out/pmc_sp8/timing.txt
for v in step4 sp8_phase sp8_fine; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sp8/$v.p1 -o p1 -- tools/micro/mfma_fillers $v > gpurun_out/pmc_sp8/$v.p1.log 2>&1 || { echo "p1 $v failed"; exit 1; }
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/pmc_sp8/$v.p2 -o p2 -- tools/micro/mfma_fillers $v > gpurun_out/pmc_sp8/$v.p2.log 2>&1 || { echo "p2 $v failed"; exit 1; }
done
python3 tools/pmc_sp8_summary.py gpurun_out/pmc_sp8 > gpurun_out/pmc_sp8/summary.txt
cat gpurun_out/pmc_sp8/summary.txt
