cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -2; }
run SDRHIP_SMALL_CHAIN=0
run SDRHIP_SMALL_CHAIN=1 SDRHIP_SMALL_CHAIN_TILE=48
run SDRHIP_SMALL_CHAIN=1 SDRHIP_SMALL_CHAIN_TILE=159 SDRHIP_STREAM_SLOTS=3
