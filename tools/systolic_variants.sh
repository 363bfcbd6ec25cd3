#!/bin/bash
# Same-box A/B of build variants of the systolic decimator (kernels_systolic.hip; SDRHIP_SYSTOLIC_DEFS): rebuild on the box, bit checks,
# then the chain's stage times (u8 kernel) and the cfloat launch (tools/k2c_seam_cost.py).  Usage: tools/systolic_variants.sh "<defs>" ...
cd ${GRAFT_REPO_ROOT:-.}
for defs in "$@"; do
  echo "=== variant: [$defs]"
  touch sdr_amd/csrc/kernels_systolic.hip
  SDRHIP_SYSTOLIC_DEFS="$defs" python -m sdr_amd.build 2>&1 | grep -v "^/" | tail -3
  timeout 600 python -m pytest tests/test_gpu_systolic.py -m gpu -x -q 2>&1 | tail -1
  timeout 300 python tools/k2k3_fusion_ab.py resamp_demod 2>&1 | grep "^fusion 1"
  timeout 300 python tools/k2c_seam_cost.py 2>&1 | grep "cfloat /8" | tail -2
done
