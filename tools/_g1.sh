cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
python bench.py 2>gpurun_out/r03g/bench_full.err > gpurun_out/r03g/bench_full.json
tail -c 600 gpurun_out/r03g/bench_full.json
tools/k2lab/lab 27 50 2 > gpurun_out/r03g/k2lab_run5.txt 2>&1
tools/k2lab/small_lab 20 > gpurun_out/r03g/small_lab.txt 2>&1
tools/k2lab/small_lab 20 0 >> gpurun_out/r03g/small_lab.txt 2>&1
tools/k2lab/small_lab 17 >> gpurun_out/r03g/small_lab.txt 2>&1
tools/k2lab/small_lab 21 >> gpurun_out/r03g/small_lab.txt 2>&1
python tools/stage_bench.py 26 2>&1 | grep -E "fmDemod|resample|filter" > gpurun_out/r03g/stage_bench.txt
python tools/host_stream_native.py > gpurun_out/r03g/host_stream.txt 2>&1
tail -5 gpurun_out/r03g/host_stream.txt
