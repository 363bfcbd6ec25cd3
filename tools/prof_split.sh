R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/prof_split.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/pmc_sq -o p -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/pmc_lds -o p -- $CMD > $OUT/lds.log 2>&1
python $R/tools/pmc_summary.py $(find $OUT -name "*.db")
