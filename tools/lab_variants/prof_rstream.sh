#!/bin/bash
# rocprofv3 of tools/prof_rstream.py: kernel-trace stats, then PMC passes (their own runs).  Output: gpurun_out/prof_rstream_<tag>/
TAG=${1:-a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_rstream_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/prof_rstream.py 20"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats_run.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/sq_run.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_lds -o p -- $CMD > $OUT/lds_run.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_misc -o p -- $CMD > $OUT/misc_run.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/fetch_run.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/write_run.log 2>&1
cd $R
python tools/pmc_csv_summary.py $OUT resample3

