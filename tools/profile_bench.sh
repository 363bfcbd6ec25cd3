#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + HBM PMC counters of the bench command.
# Counters are collected in their own passes, with --kernel-trace/--stats only (gpurun refuses --pmc
# combined with sys/hip/hsa/memory-copy traces).  Outputs land in gpurun_out/prof_<tag>/.
TAG=${1:-r01}
BLOCKS=${2:-65536}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="env BENCH_NO_LIB_OVERLAP=1 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --passes-per-step 1 --blocks $BLOCKS"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/bench_stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > $OUT/bench_fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $CMD > $OUT/bench_write_run.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > $OUT/bench_sq_run.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o bench -- $CMD > $OUT/bench_lds_run.log 2>&1
# the cfloat-in instantiation of the decimator (BASELINE configs[1]) on its own
K2C="python $R/tools/prof_k2.py 27 f32 8192 400 200"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_k2c -o bench -- $K2C > $OUT/k2c_fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_k2c -o bench -- $K2C > $OUT/k2c_write_run.log 2>&1
# round 3: kernel-trace + clock of the cfloat-in instantiation (the north_star kernel, BASELINE configs[1]) on its own ...
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_k2c -o bench -- $K2C > $OUT/k2c_stats_run.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq_k2c -o bench -- $K2C > $OUT/k2c_sq_run.log 2>&1
# ... and of the one-kernel chain at BASELINE configs[4]'s shard size (2^20 samples per pass)
SHARD="python $R/tools/shard_pass_probe.py 128 400"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shard -o bench -- $SHARD > $OUT/shard_stats_run.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmc_shard -o bench -- $SHARD > $OUT/shard_pmc_run.log 2>&1
# un-profiled reference run of the same command
$CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
find $OUT -name "*.csv" | head -50
