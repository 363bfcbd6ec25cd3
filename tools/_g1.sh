cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for w in 0 1; do echo -n "NO_PREFETCH=$w: "; SDRHIP_FIR_NO_PREFETCH=$w python tools/stage_bench.py 26 2>&1 | grep -E "filter.*8192"; done; done
