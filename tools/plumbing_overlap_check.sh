cd $GRAFT_REPO_ROOT
export BENCH_TRANSPORT=host BENCH_CHECKSUM=1
for m in 0 1; do
BENCH_NO_OVERLAP=$m timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --blocks 4096 --no-cpu-baseline --no-extras 2>&1 | grep -o '"value[^,]*,\|"audio_crc32_per_rank[^]]*]\|"sharding[^}]*' 
done
python -m pytest tests/test_gpu_chain.py -q 2>&1 | tail -2
