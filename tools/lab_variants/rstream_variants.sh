#!/bin/bash
# Same-box A/B of build variants of the fused fmDemod + resampler kernels: for each "<stream defs>|<tile-kernel defs>" rebuild
# kernels_resample_stream.hip / kernels_chain.hip on the box, check bits (tests/test_gpu_resample_stream.py and the fusion tests of
# tests/test_gpu_fullsize.py), then tools/k2k3_fusion_ab.py resamp_stream (alternating stream (fusion 1) / tile kernel (fusion 0) in
# one process, per-stage HIP-event times).  Usage: tools/rstream_variants.sh "<defs>|<defs>" ...   (SKIP_TESTS=1: timing only)
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  sdefs="${v%%|*}"; cdefs=""; [[ "$v" == *"|"* ]] && cdefs="${v#*|}"
  echo "=== variant: stream [$sdefs] tile [$cdefs]"
  touch sdr_amd/csrc/kernels_resample_stream.hip sdr_amd/csrc/kernels_chain.hip
  SDRHIP_RSTREAM_DEFS="$sdefs" SDRHIP_CHAIN_DEFS="$cdefs" python -m sdr_amd.build 2>&1 | grep -v "^/" | tail -3
  if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_resample_stream.py "tests/test_gpu_fullsize.py::test_chain_demod_fusion_is_invisible" -m gpu -x -q 2>&1 | tail -4; fi
  timeout 400 python tools/k2k3_fusion_ab.py resamp_stream 2>&1 | grep "^fusion"
done
