#!/bin/bash
# round 3: small-map convolution kernels, A/B of the dispatch switches in one box visit
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for m in "0 0" "1 1" "2 2"; do
  set -- $m
  CGAMD_SCONV=$1 CGAMD_SWGRAD=$2 timeout 300 python scripts/check_small_conv.py > gpurun_out/r3b_s$1w$2.txt 2>&1
  cat gpurun_out/r3b_s$1w$2.txt | grep -v amdgpu.ids
done
