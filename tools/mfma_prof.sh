cd /tmp && export TMPDIR=/tmp
export COAST_MM_ENGINE=mfma
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_INSTS_VALU|SQ_ACTIVE_INST|SQ_WAIT|LDS_BANK|SQ_INSTS_LDS|SQ_BUSY_CY|SQ_WAVE_CYCLES" | head -40 > $R/gpurun_out/counters.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/prof_mfma/pmc_sq -o bench -- python $R/tools/perf_kernels.py --only mm > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_mfma/pmc_lds -o bench -- python $R/tools/perf_kernels.py --only mm > /dev/null 2>&1
python $R/tools/summarize_prof.py $R/gpurun_out/prof_mfma | grep -E "mfma.*kernel<3>"
find $R/gpurun_out/prof_mfma -name "*.db" -delete
