cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
export BENCH_TRANSPORT=host
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 4096 2>gpurun_out/r03e/bench2.err | tee gpurun_out/r03e/bench2.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}); print(d['config']); print('shard',d['shard_1M_samples_per_gpu']); print('noex',d['without_halo_exchange'])"
tail -5 gpurun_out/r03e/bench2.err
unset BENCH_TRANSPORT
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 1 --blocks 4096 --no-extras 2>gpurun_out/r03e/bench2r.err | tee gpurun_out/r03e/bench2r.json | cut -c1-900
tail -5 gpurun_out/r03e/bench2r.err
