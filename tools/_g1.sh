cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -8
python tools/shard_pass_probe.py 128 2>&1 | grep blocks | tee gpurun_out/r03d/probe128.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/r03d/bench.err | tee gpurun_out/r03d/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'ms_per_step',d['ms_per_step']); print('shard',d['shard_1M_samples_per_gpu']); print('host',d['host_streamed']); print('stage',d['stage_ms'])"
